"""One gloo rank of the CPU multi-process host-logic test (launched by test_host_logic.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    import torch.distributed as dist
    rank, size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    cfg = json.loads(sys.argv[1])
    outdir = sys.argv[2]
    dist.init_process_group("gloo", rank=rank, world_size=size)
    import cpu_world
    from offt_amd import api
    for ci, case in enumerate(cfg):
        p1 = case["params"].get("P1")
        if p1 is None:
            import oracle_lib as O
            p1 = O.params_default(case["N"][0], case["N"][1], case["N"][2], size)[0]
        cpu_world.install(rank, size, p1=p1, dist=dist)
        res = cpu_world.run_rank(*case["N"], kind=1, is_equalxy=case.get("eq", 0), is_r2c=case.get("r2c", 0),
                                 roundtrip=bool(case.get("inv")), **case["params"])
        c, v, buf = res[:3]
        np.save(os.path.join(outdir, f"case{ci}_rank{rank}.npy"), buf)
        if case.get("inv"):
            np.save(os.path.join(outdir, f"case{ci}_rank{rank}_inv.npy"), cpu_world.input_block(c, res[3]))
        json.dump({"comm": c, "v": v}, open(os.path.join(outdir, f"case{ci}_rank{rank}.json"), "w"))
        dist.barrier()
    cpu_world.uninstall()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
