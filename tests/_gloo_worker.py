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
        for k, v in case.get("env", {}).items():
            os.environ[k] = str(v).replace("{outdir}", outdir)
        cpu_world.install(rank, size, dist=dist)
        res = cpu_world.run_rank(*case["N"], kind=1, is_equalxy=case.get("eq", 0), is_r2c=case.get("r2c", 0),
                                 roundtrip=bool(case.get("inv")), max_loop=case.get("max_loop", 0), **case["params"])
        for k in case.get("env", {}):
            os.environ.pop(k, None)
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
