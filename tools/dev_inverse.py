#!/usr/bin/env python3
"""Developer perf probe: forward against inverse on one GPU, per pass.   tools/dev_inverse.py N [f64|f32] [reps]"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    prec = api.F32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else api.F64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    po = api.offt_3d_init(n, n, n, precision=prec)
    L = api.lib()
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
    esz = 16 if prec == api.F64 else 8
    E = n ** 3
    for name, d in (("forward", -1), ("inverse", +1)):
        best = None
        for _ in range(reps):
            L.offt_hip_fill_input(po, dev.data_ptr(), 1)
            if d > 0:
                api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), d)
            t = (C.c_double * 3)()
            L.offt_hip_last_pass_seconds(po, t)
            tot = L.offt_hip_last_device_seconds(po)
            if best is None or tot < best[0]:
                best = (tot, list(t))
        parts = " ".join(f"{'zyx'[i]} {best[1][i]*1e3:.3f}ms {2*esz*E/best[1][i]/8e12*100:.1f}%" for i in range(3) if best[1][i] > 0)
        print(f"{n}^3 {'f64' if prec == api.F64 else 'f32'} {name}: {best[0]*1e3:.3f} ms ({6*esz*E/best[0]/8e12*100:.1f}% of 8 TB/s)  {parts}", flush=True)
    api.offt_3d_fin(po)


if __name__ == "__main__":
    main()
