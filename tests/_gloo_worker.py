"""One gloo rank of the CPU multi-process host-logic test (launched by test_host_logic.py)."""
import ctypes as C
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
        CB = cpu_world.install(rank, size, dist=dist)
        fa = case.get("fail_alloc")
        if fa:
            # one rank runs out of memory while the static sweep rebuilds the mesh for its SECOND point: count what a plan
            # without the sweep allocates (a0), then let allocation a0 + 2 fail (a0 at init, 1 sweep scratch array, then the
            # first buffer of the rebuilt mesh).  Every rank makes the dry plan: init is collective.
            CB.cpu_backend_alloc_count.restype = C.c_long
            CB.cpu_backend_fail_alloc_at.argtypes = [C.c_long]
            n0 = CB.cpu_backend_alloc_count()
            dry = api.offt_3d_init(*case["N"], custom_params=api.make_params(**case["params"]))
            a0 = CB.cpu_backend_alloc_count() - n0
            api.offt_3d_fin(dry)
            if rank == fa["rank"]:
                CB.cpu_backend_fail_alloc_at(a0 + 2 + fa.get("extra", 0))
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
