#!/usr/bin/env python3
"""Developer probe: real-to-complex (-R) transform time and pass split."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api
L = api.lib()
for n in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    po = api.offt_3d_init(n, n, n, is_r2c=1)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(4):
        L.offt_hip_fill_input(po, dev.data_ptr(), 1)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        t = (C.c_double * 3)(); L.offt_hip_last_pass_seconds(po, t)
        d = L.offt_hip_last_device_seconds(po)
        if best is None or d < best[0]: best = (d, list(t))
    h = n // 2 + 1
    alg = [8.0 * n ** 3 + 16.0 * h * n * n, 32.0 * h * n * n, 32.0 * h * n * n]
    print(f"r2c {n}^3: {best[0]*1e3:.3f} ms; z/y/x " + " ".join(f"{x*1e3:.3f}ms({a/x/1e9:.0f}GB/s)" for x, a in zip(best[1], alg))
          + f" => {sum(alg)/best[0]/8e12*100:.1f}% of 8 TB/s", flush=True)
    api.offt_3d_fin(po)
