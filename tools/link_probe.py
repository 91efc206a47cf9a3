#!/usr/bin/env python3
"""xGMI link microbenchmark (hard part #1: decide mesh shapes with a MEASURED link number).

  python tools/link_probe.py --gpus N [--sizes-mib 8,32,128,256] [--reps 5]

One process per GPU (self-launching like bench.py, or under torch.distributed.run).  Times the exact primitive of the
FFT's exchanges -- grouped ncclSend/ncclRecv on the library's world communicator (offt_hip_link_probe) --
  * pairwise: ring shift by d = 1 .. N/2 (every GPU sends to rank+d and receives from rank-d: one link per direction),
  * all-to-all among all N GPUs (N-1 links per GPU at once),
and prints GB/s per link and direction, max time over ranks, next to the 153 GB/s (bidirectional) per-link figure.
Rank 0 prints one JSON line per measurement and a summary line."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # launcher only (no torch import at module level)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--sizes-mib", default="8,32,128,256")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--launch-timeout", type=float, default=600.0)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(bench.launch(args, sys.argv[1:], script=__file__))
    import torch
    import torch.distributed as dist
    from offt_amd import api
    rank, world, lr = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    torch.cuda.set_device(lr)
    L = api.lib()
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_char * 128)()
        assert L.offt_hip_get_unique_id(buf) == 0, L.offt_hip_last_error()
        uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    uid = uid.cuda()
    dist.broadcast(uid, src=0)
    assert L.offt_hip_set_world(rank, world, bytes(uid.cpu().numpy().tobytes()), lr) == 0, L.offt_hip_last_error()

    def mx(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    rows = []
    for mib in [int(x) for x in args.sizes_mib.split(",")]:
        nbytes = mib << 20
        for d in range(1, world // 2 + 1):
            dist.barrier()
            sec = mx(L.offt_hip_link_probe(1, d, nbytes, args.reps))
            rows.append({"pattern": f"shift {d}", "MiB_per_peer": mib, "seconds": sec, "GBps_per_link_per_direction": nbytes / sec / 1e9 if sec > 0 else None})
        if world > 1:
            dist.barrier()
            sec = mx(L.offt_hip_link_probe(0, 0, nbytes, args.reps))
            rows.append({"pattern": f"all-to-all x{world}", "MiB_per_peer": mib, "seconds": sec,
                         "GBps_per_link_per_direction": nbytes / sec / 1e9 if sec > 0 else None,
                         "GBps_out_of_each_gpu": (world - 1) * nbytes / sec / 1e9 if sec > 0 else None})
    if rank == 0:
        for r in rows:
            print(json.dumps(r), flush=True)
        best = max((r["GBps_per_link_per_direction"] or 0) for r in rows) if rows else 0
        print(json.dumps({"summary": True, "n_gpus": world, "ranks_seen": L.offt_hip_world_count(),
                          "best_GBps_per_link_per_direction": best, "link_spec_GBps_bidirectional": 153.0}), flush=True)
    dist.barrier()
    L.offt_hip_finalize_world()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
