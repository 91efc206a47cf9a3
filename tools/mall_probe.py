#!/usr/bin/env python3
"""Developer probe: does the x pass find the y pass's output in the Infinity Cache (256 MiB, memory side) when the two
passes of the single-GPU z-y-x schedule alternate over groups of z-planes instead of running one after the other?
  tools/mall_probe.py [N] [G ...]      per G: device time of  y(all); x(all)  against  for g: y(g); x(g)
Under rocprofv3 --pmc FETCH_SIZE the per-launch HBM reads tell the hit rate."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from offt_amd import api
from test_gpu_descriptors import Desc


KEEP = [0]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    groups = [int(x) for x in sys.argv[2:]] or [4, 8, 16, 32, 128]
    L = api.lib()
    L.offt_hipk_fft_pass.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.offt_hipk_prepare.argtypes = [C.c_int, C.c_int]
    assert L.offt_hipk_prepare(N, api.F64) == 0
    wy, wx = N, N * N + 72
    data = torch.zeros(N * N * N * 2, dtype=torch.float64, device="cuda")
    W = torch.ones(N * wx * 2, dtype=torch.float64, device="cuda")
    os0, os1, os2 = 1, N, N * N

    def ypass(z0, G):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = N, api.F64, -1, N, G, 1
        d.in_axis_stride, d.in_col_stride, d.in_b1_stride, d.in_contig = 1, wx, wy, 1
        d.out_axis_stride, d.out_col_stride, d.out_b1_stride, d.out_contig = os1, os0, os2, 0
        d.variant, d.scale, d.out_keep = -1, 1.0 / N, KEEP[0]
        assert L.offt_hipk_fft_pass(C.byref(d), W.data_ptr() + 16 * z0 * wy, data.data_ptr() + 16 * z0 * os2, None) == 0

    def xpass(z0, G):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = N, api.F64, -1, N, G, 1
        d.in_axis_stride = d.out_axis_stride = os0
        d.in_col_stride = d.out_col_stride = os1
        d.in_b1_stride = d.out_b1_stride = os2
        d.in_contig = d.out_contig = 1
        d.variant, d.scale = -1, 1.0 / N
        p = data.data_ptr() + 16 * z0 * os2
        assert L.offt_hipk_fft_pass(C.byref(d), p, p, None) == 0

    # out-of-place variant: y(g) -> ring slot R[g % 2] (kept stores), x(g): ring slot -> data
    ring = {}

    def ypass_ring(z0, G, slot, keep=1):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = N, api.F64, -1, N, G, 1
        d.in_axis_stride, d.in_col_stride, d.in_b1_stride, d.in_contig = 1, wx, wy, 1
        d.out_axis_stride, d.out_col_stride, d.out_b1_stride, d.out_contig = os1, os0, os2, 0
        d.variant, d.scale, d.out_keep = -1, 1.0 / N, keep
        assert L.offt_hipk_fft_pass(C.byref(d), W.data_ptr() + 16 * z0 * wy, ring[G].data_ptr() + 16 * slot * G * os2, None) == 0

    def xpass_ring(z0, G, slot):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = N, api.F64, -1, N, G, 1
        d.in_axis_stride = d.out_axis_stride = os0
        d.in_col_stride = d.out_col_stride = os1
        d.in_b1_stride = d.out_b1_stride = os2
        d.in_contig = d.out_contig = 1
        d.variant, d.scale = -1, 1.0 / N
        assert L.offt_hipk_fft_pass(C.byref(d), ring[G].data_ptr() + 16 * slot * G * os2, data.data_ptr() + 16 * z0 * os2, None) == 0

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    KEEP[0] = 0
    t_all = timed(lambda: (ypass(0, N), xpass(0, N)))
    KEEP[0] = 1
    print(f"{N}^3 f64  y(all); x(all): {t_all:.3f} ms", flush=True)
    for G in groups:
        def alt():
            for z0 in range(0, N, G):
                ypass(z0, G)
                xpass(z0, G)
        t = timed(alt)
        def seq():
            for z0 in range(0, N, G):
                ypass(z0, G)
            for z0 in range(0, N, G):
                xpass(z0, G)
        t2 = timed(seq)
        ring[G] = torch.zeros(2 * G * N * N * 2, dtype=torch.float64, device="cuda")
        def oop():
            for k, z0 in enumerate(range(0, N, G)):
                ypass_ring(z0, G, k & 1)
                xpass_ring(z0, G, k & 1)
        t3 = timed(oop)
        def oop_nt():
            for k, z0 in enumerate(range(0, N, G)):
                ypass_ring(z0, G, k & 1, keep=0)
                xpass_ring(z0, G, k & 1)
        t4 = timed(oop_nt)
        del ring[G]
        print(f"  groups of {G:4d} planes ({G * N * N * 16 >> 20} MiB): alternating in place {t:.3f} ms, same launches pass after pass {t2:.3f} ms, "
              f"through a 2-slot ring (x pass out of place) {t3:.3f} ms, ring with streaming y stores {t4:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
