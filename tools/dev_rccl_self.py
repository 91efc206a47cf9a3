#!/usr/bin/env python3
"""Developer check: full-size 1024^3 through the slab schedule with every message going through RCCL
(one-rank communicator, send/recv to self), closed-form ramp verification."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OFFT_FORCE_PIPELINE"] = "1"; os.environ["OFFT_FORCE_A2A"] = "1"
import numpy as np, torch
from offt_amd import api
L = api.lib()
uid = (C.c_char * 128)()
assert L.offt_hip_get_unique_id(uid) == 0
assert L.offt_hip_set_world(0, 1, uid, 0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
po = api.offt_3d_init(n, n, n)
c = api.comm_dict(po)
dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
for rep in range(3):
    L.offt_hip_fill_input(po, dev.data_ptr(), 0)
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    print("rep", rep, "device ms", L.offt_hip_last_device_seconds(po) * 1e3, flush=True)
cv = torch.view_as_complex(dev.view(-1, 2))
os0, os1, os2 = c["ostride"]
assert complex(cv[0]) == n ** 3 * 111 * (n - 1) / 2, complex(cv[0])
for k in (1, 2, 3, n // 2 - 1):
    cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n))
    for st, f in ((os2, 1), (os1, 10), (os0, 100)):
        assert abs(complex(cv[k * st]) - f * cf) / abs(f * cf) < 1e-12
print("closed-form check ok")
api.offt_3d_fin(po); L.offt_hip_finalize_world()
