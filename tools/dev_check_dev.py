#!/usr/bin/env python3
"""Developer smoke for a dev-registry build (OFFT_AMD_LIB=build/dev/<name>/liboffthip.so): GPU vs numpy on shapes that
use the 256 / 1024 / 2048 kernels in every flavour, forward and inverse, f64 and f32, plus split addressing through the
forced tile pipeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from dev_gpu_check import run
from offt_amd import api
bad = 0
tol = {api.F64: 1e-13, api.F32: 5e-6}
for prec in (api.F64, api.F32):
    for shape in ((2048, 8, 8), (8, 2048, 8), (8, 8, 2048), (1024, 16, 8), (8, 1024, 16), (16, 8, 1024), (256, 256, 16), (2048, 2048, 8)):
        for (S, eq) in ((1, 0), (0, 0), (0, 1)):
            if eq and shape[0] != shape[1]:
                continue
            bad += run(*shape, S=S, eq=eq, prec=prec, inverse=True) > tol[prec]
os.environ["OFFT_FORCE_PIPELINE"] = "1"
for prec in (api.F64, api.F32):
    for shape in ((2048, 16, 8), (16, 2048, 8), (8, 16, 2048), (1024, 1024, 8)):
        bad += run(*shape, prec=prec, inverse=True) > tol[prec]
        bad += run(*shape, S=1, prec=prec, inverse=True) > tol[prec]
print("FAILED" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
