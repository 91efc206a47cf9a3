#!/usr/bin/env python3
"""Developer probe: forced tile pipeline on one GPU for shapes that mimic a rank's share of a multi-GPU run."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api
os.environ["OFFT_FORCE_PIPELINE"] = "1"
def run(shape, slab, T1=None, reps=4):
    os.environ["OFFT_NO_SLAB_LAYOUT"] = "0" if slab else "1"
    kw = {} if T1 is None else dict(T1=T1)
    po = api.offt_3d_init(*shape, custom_params=api.make_params(**kw))
    L = api.lib()
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
    best = None
    for r in range(reps):
        L.offt_hip_fill_input(po, dev.data_ptr(), 1)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        t = (C.c_double * 3)(); L.offt_hip_last_pass_seconds(po, t)
        tot = L.offt_hip_last_device_seconds(po)
        if best is None or tot < best[0]: best = (tot, list(t))
    E = shape[0] * shape[1] * shape[2]
    print(f"{shape} slab={slab} T1={T1}: total {best[0]*1e3:.3f} ms tiled {best[1][0]*1e3:.3f} K3 {best[1][2]*1e3:.3f}  alg {6*16*E/best[0]/1e9:.0f} GB/s", flush=True)
    api.offt_3d_fin(po)
for shape in ((1024, 128, 1024), (1024, 1024, 128)):
    for rep in range(2):
        for T1 in (None, 64, 128, 256, 512):
            run(shape, 1, T1=T1)
    run(shape, 0)
