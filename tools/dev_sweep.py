#!/usr/bin/env python3
"""Developer probe: time every registered 1024 variant on the z-y-x schedule (OFFT_AMD_LIB selects the build)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from offt_amd import api
from tools.dev_perf import run
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PREC = api.F32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else api.F64
L = api.lib()
nv = L.offt_hipk_variant_count(N, PREC)
for rep in range(2):
    for v in range(nv):
        print("variant", v, L.offt_hipk_variant_name(N, PREC, v).decode(), flush=True)
        try:
            run(N, 0, 0, (v, v, v), prec=PREC, reps=3 if N > 1024 else 5)
        except Exception as e:
            print("  failed:", e)
