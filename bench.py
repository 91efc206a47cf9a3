#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native OFFT hot path.

Metric (BASELINE.json): 3-D FFT GFLOP/s + fraction of the HBM roofline, 1024^3
double-complex, forward transform (offt_3d_execute), data resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A "step" is one offt_3d_execute of the whole grid.  flops = 5 E log2 E,
algorithmic bytes = 6 * 16 B * E / P per GPU per transform.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable copy)


def cpu_baseline(n, cores):
    """The oracle (CPU restatement of the reference pipeline: default parameters, tiles, pack ->
    all-to-all -> unpack, 1-D FFTs) timed on this host: one simulated rank per core as OpenMP
    threads.  Checker infrastructure used here ONLY as the reported CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    p = 1
    while p * 2 <= cores:
        p *= 2
    w = L.orc_world_create(n, n, n, p, 0, 0, 0, None)
    L.orc_world_fill(w, 1)
    L.orc_world_execute(w, p)  # warm-up: first touch of the exchange buffers (the GPU side is warmed up too)
    dt = None
    for _ in range(2):         # min of reps, like the reference harness (run-fft.c:408-413)
        L.orc_world_fill(w, 1)
        t0 = time.time()
        L.orc_world_execute(w, p)
        d = time.time() - t0
        dt = d if dt is None else min(dt, d)
    L.orc_world_destroy(w)
    flops = 5.0 * n ** 3 * math.log2(n ** 3)
    return {"value": round(flops / dt / 1e9, 3), "unit": "GFLOP/s", "cores": p, "kind": "port",
            "sample": f"{n}^3 double-complex forward, best of 2 after 1 warm-up, {p} simulated MPI ranks (one OpenMP thread each), "
                      f"reference default parameters, {dt:.2f} s per transform"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1024, help="grid side (default 1024: BASELINE configs[2])")
    ap.add_argument("--p1", type=int, default=-1, help="mesh rows; default 1 (one exchange over all xGMI links)")
    ap.add_argument("--layout", default="zyx", choices=["zyx", "xyz"], help="output layout: reference default z-y-x, or S=1 x-y-z")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=512)
    args = ap.parse_args()

    import torch
    from offt_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    L = api.lib()
    dist = None
    force_dist = bool(int(os.environ.get("OFFT_BENCH_FORCE_DIST", "0")))  # rehearse the multi-rank plumbing on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            if L.offt_hip_get_unique_id(buf):
                raise SystemExit("offt_hip_get_unique_id failed: " + L.offt_hip_last_error().decode())
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        uid = uid.cuda()
        dist.broadcast(uid, src=0)
        idb = bytes(uid.cpu().numpy().tobytes())
        if L.offt_hip_set_world(rank, world, idb, local_rank):
            raise SystemExit("offt_hip_set_world failed: " + L.offt_hip_last_error().decode())

    n = args.n
    E = float(n) ** 3
    params = {}
    if world > 1 or force_dist:
        params["P1"] = args.p1 if args.p1 > 0 else 1
    if args.layout == "xyz":
        params["S"] = 1
    po = api.offt_3d_init(n, n, n, custom_params=api.make_params(**params))
    c = api.comm_dict(po)
    nel = api.local_elems(po)
    data = torch.zeros(nel * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # the plan owns a non-blocking stream: order it after torch's zero fill
    L.offt_hip_fill_input(po, data.data_ptr(), 1)  # seeded position hash in [-1, 1)
    # keep magnitudes bounded over many back-to-back transforms: exact power-of-two rescale in the last store
    L.offt_hip_set_output_scale(po, 2.0 ** -(round(math.log2(E)) // 2))
    ptr = data.data_ptr()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        api.offt_3d_execute(po, ptr, ptr)
    barrier()
    pass_acc = [0.0, 0.0, 0.0]
    dev_acc = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        api.offt_3d_execute(po, ptr, ptr)  # returns after the GPU finished; per-pass HIP events inside
        t3 = (C.c_double * 3)()
        L.offt_hip_last_pass_seconds(po, t3)
        for i in range(3):
            pass_acc[i] += t3[i]
        dev_acc += L.offt_hip_last_device_seconds(po)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    api.offt_3d_fin(po)

    if rank == 0:
        steps = args.steps
        flops = 5.0 * E * math.log2(E)
        ms = dt / steps * 1e3
        value = flops * steps / dt / 1e9
        alg_bytes_transform = 6.0 * 16.0 * E / world
        # dominant kernel = the slowest of the three panel-FFT launches of one transform; one launch
        # reads and writes every local element once: 2 * 16 B * E / P algorithmic bytes
        names = ["z", "y", "x"]
        if world == 1:
            k = max(range(3), key=lambda i: pass_acc[i])
            kdur = pass_acc[k] / steps
            kname = f"fft_panel_k ({names[k]}-axis pass)"
            alg_launch = 2.0 * 16.0 * E
        else:
            # multi-rank slab schedule: phase 0 = the FFTz launches (K1, all x-tiles, no waiting on the
            # exchange); phase 2 = z-chunks of exchange-wait + FFTy + FFTx, which includes time on the wire
            k = 0
            kdur = pass_acc[0] / steps
            kname = "fft_panel_k (z-axis pass K1, all x-tiles)"
            alg_launch = 2.0 * 16.0 * E / world
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj) and world == 1 and n == 1024:
            try:
                traffic = json.load(open(tj)).get(names[k] + "_pass_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(alg_launch / kdur / 1e9, 1) if kdur > 0 else None,
                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(alg_launch / kdur / HBM_PEAK, 4) if kdur > 0 else None,
                "traffic": traffic, "avg_launch_ms": round(kdur * 1e3, 4),
                "alg_bytes_per_launch": alg_launch,
                "pass_ms": {names[i]: round(pass_acc[i] / steps * 1e3, 4) for i in range(3)},
                "transform_frac": round(alg_bytes_transform / (dt / steps) / HBM_PEAK, 4),
                "transform_device_ms": round(dev_acc / steps * 1e3, 4)}
        out = {"metric": "3D FFT GFLOP/s (1024^3 double-complex forward, 5*E*log2(E) flop model)" if n == 1024 else f"3D FFT GFLOP/s ({n}^3 double-complex forward)",
               "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic (seeded position hash, device-resident)",
               "config": {"workload": f"{n}^3 double-complex forward 3-D FFT, in-place, offt_3d_execute",
                          "grid": [n, n, n], "mesh": f"{c['p1']}x{c['p2']}", "output_layout": args.layout,
                          "tile_T1": "reference default (M1/16)"},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
                out["cpu_baseline"] = cpu_baseline(args.cpu_n, min(cores, 16))
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        L.offt_hip_finalize_world()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
