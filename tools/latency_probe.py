#!/usr/bin/env python3
"""Developer probe: wall time per offt_3d_execute for small grids (launch-bound regime) vs device time."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api
L = api.lib()
for n in (32, 64, 128, 256):
    po = api.offt_3d_init(n, n, n)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    L.offt_hip_set_output_scale(po, 1.0 / n ** 1.5)
    for _ in range(20):
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    dt = (time.perf_counter() - t0) / reps
    d = L.offt_hip_last_device_seconds(po)
    t = (C.c_double * 3)(); L.offt_hip_last_pass_seconds(po, t)
    # asynchronous mode: enqueue only, one sync at the end
    L.offt_hip_set_async(po, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    L.offt_hip_device_synchronize()
    da = (time.perf_counter() - t0) / reps
    print(f"{n}^3: sync execute {dt*1e6:.1f} us wall, device {d*1e6:.1f} us (passes {t[0]*1e6:.1f}/{t[1]*1e6:.1f}/{t[2]*1e6:.1f}); async back-to-back {da*1e6:.1f} us per transform", flush=True)
    api.offt_3d_fin(po)
