#!/usr/bin/env python3
"""Developer perf probe for one (possibly non-cubic) grid on one GPU: per-pass device time and fraction of the
HBM roofline.   tools/shape_probe.py Nx,Ny,Nz [f64|f32] [S] [reps] [vx,vy,vz] [eq]
Each pass moves 2 * esz * E bytes; the pass of axis length n is reported with its kernel name."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api


def run(shape, prec=api.F64, S=0, reps=4, variants=(-1, -1, -1), eq=0):
    po = api.offt_3d_init(*shape, custom_params=api.make_params(S=S), is_equalxy=eq, precision=prec)
    L = api.lib()
    for ax, v in enumerate(variants):
        L.offt_hip_set_variant(po, ax, v)
    n = api.local_elems(po)
    dev = torch.zeros(n * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        L.offt_hip_fill_input(po, dev.data_ptr(), 1)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        t = (C.c_double * 3)()
        L.offt_hip_last_pass_seconds(po, t)
        tot = L.offt_hip_last_device_seconds(po)
        if best is None or tot < best[0]:
            best = (tot, list(t))
    esz = 16 if prec == api.F64 else 8
    E = shape[0] * shape[1] * shape[2]
    names = "zyx"
    dims = (shape[2], shape[1], shape[0])
    parts = " ".join(f"{names[i]}(n={dims[i]}) {best[1][i]*1e3:.3f}ms {2*esz*E/best[1][i]/8e12*100:.1f}%" for i in range(3) if best[1][i] > 0)
    print(f"{shape} {'f64' if prec == api.F64 else 'f32'} S={S} eq={eq} var={variants}: total {best[0]*1e3:.3f} ms "
          f"({6*esz*E/best[0]/8e12*100:.1f}% of 8 TB/s, {5*E*math.log2(E)/best[0]/1e9:.0f} GFLOP/s)  {parts}", flush=True)
    api.offt_3d_fin(po)
    del dev


if __name__ == "__main__":
    shape = tuple(int(x) for x in sys.argv[1].split(","))
    prec = api.F32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else api.F64
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    var = tuple(int(x) for x in sys.argv[5].split(",")) if len(sys.argv) > 5 else (-1, -1, -1)
    eq = int(sys.argv[6]) if len(sys.argv) > 6 else 0   # 1: is_equalxy (y-z-x output, Nx == Ny)
    run(shape, prec, S, reps, var, eq)
