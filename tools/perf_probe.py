#!/usr/bin/env python3
"""Developer perf probe: per-pass device time for one grid, all layouts/variants."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api

def run(N, S, eq, variants=(-1, -1, -1), prec=api.F64, reps=4):
    cp = api.make_params(S=S)
    po = api.offt_3d_init(N, N, N, custom_params=cp, is_equalxy=eq, precision=prec)
    L = api.lib()
    for ax, v in enumerate(variants):
        L.offt_hip_set_variant(po, ax, v)
    n = api.local_elems(po)
    dev = torch.zeros(n * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    best = None
    for r in range(reps):
        L.offt_hip_fill_input(po, dev.data_ptr(), 1)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        t = (C.c_double * 3)(); L.offt_hip_last_pass_seconds(po, t)
        tot = L.offt_hip_last_device_seconds(po)
        if best is None or tot < best[0]: best = (tot, list(t))
    esz = 16 if prec == api.F64 else 8
    E = N ** 3
    gbs = [2 * esz * E / x / 1e9 if x > 0 else 0 for x in best[1]]
    import math
    print(f"N={N} S={S} eq={eq} var={variants} prec={prec}: total {best[0]*1e3:.3f} ms  z/y/x = "
          + " ".join(f"{x*1e3:.3f}ms({g:.0f}GB/s)" for x, g in zip(best[1], gbs))
          + f"  => {5*E*math.log2(E)/best[0]/1e9:.0f} GFLOP/s, alg {6*esz*E/best[0]/1e9:.0f} GB/s ({6*esz*E/best[0]/8e12*100:.1f}% of 8TB/s)", flush=True)
    api.offt_3d_fin(po)
    del dev

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 2 and sys.argv[2] == "f32":
        run(N, 0, 0, prec=api.F32, reps=3)
        run(N, 1, 0, prec=api.F32, reps=3)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "f64":
        run(N, 0, 0, reps=3)
        run(N, 1, 0, reps=3)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "vsweep":  # every registered variant, zyx layout
        prec = api.F32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else api.F64
        nv = api.lib().offt_hipk_variant_count(N, prec)
        for v in range(nv):
            print("variant", v, api.lib().offt_hipk_variant_name(N, prec, v).decode(), flush=True)
            run(N, 0, 0, (v, v, v), prec=prec, reps=3)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "zyx":
        for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
            run(N, 0, 0, reps=6)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "wpad":
        for pad in (0, 8, 64, 72, 520, 1032, 4104):
            os.environ["OFFT_WPAD"] = str(pad)
            print("OFFT_WPAD", pad)
            run(N, 0, 0)
        sys.exit(0)
    for S, eq in ((1, 0), (0, 0), (0, 1)):
        run(N, S, eq)
    nv = api.lib().offt_hipk_variant_count(N, 0)
    for v in range(1, nv):
        print("variant", v, api.lib().offt_hipk_variant_name(N, 0, v).decode())
        run(N, 1, 0, (v, v, v))
        run(N, 0, 0, (v, v, v))
    run(N, 1, 0, prec=api.F32)
    run(N, 0, 0, prec=api.F32)
