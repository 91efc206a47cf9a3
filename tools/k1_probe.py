#!/usr/bin/env python3
"""The K1 pass of the pencil schedule in isolation (FFTz + pack of one x-tile, offt_host.c execute_pipeline), as a bare
descriptor on offt_hipk_fft_pass, with the pitches of its send layout [peer][x_t][z_l][y] varied -- what do the 6-8 points
between K1 on the pencil meshes and a single-rank z pass of the same launch size depend on?

  python tools/k1_probe.py <p1> <p2> [rowpad planepad blkpad ...triples]      (1024^3 f64, one x-tile of M1/4 planes)
rowpad: elements added to the y-line pitch M2; planepad: to the x_t plane pitch; blkpad: to the per-peer block pitch."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from offt_amd import api
from test_gpu_descriptors import Desc


def main():
    p1, p2 = int(sys.argv[1]), int(sys.argv[2])
    pads = [int(x) for x in sys.argv[3:]] or [0, 0, 0]
    n = 1024
    M1, M2, M3 = n // p1, n // p2, n // p2
    T = M1 // 4
    L = api.lib()
    L.offt_hipk_fft_pass.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p]
    if L.offt_hipk_prepare(n, api.F64):
        raise SystemExit("prepare failed")
    NT = int(os.environ.get("K1_PROBE_TILES", "1"))  # > 1: every repetition works on another tile and another send slot (nothing of
    src = torch.randn(NT * T * M2 * n * 2, dtype=torch.float64, device="cuda")  # the previous repetition left in the caches)
    for i in range(0, len(pads), 3):
        rowpad, planepad, blkpad = pads[i:i + 3]
        row = M2 + rowpad
        plane = M3 * row + planepad
        blk = T * plane + blkpad
        dst = torch.zeros(NT * (p2 * blk * 2 + 64), dtype=torch.float64, device="cuda")
        d = Desc()
        d.n, d.precision, d.direction, d.variant, d.scale = n, api.F64, -1, -1, 1.0
        d.ncols, d.nb1, d.nb2 = M2, T, 1
        d.in_axis_stride, d.in_col_stride, d.in_b1_stride, d.in_contig = 1, n, M2 * n, 1
        d.out_axis_stride, d.out_col_stride, d.out_b1_stride, d.out_contig = row, 1, plane, 0
        if p2 > 1:
            d.out_split, d.out_block_stride = M3, blk
        BURST = int(os.environ.get("K1_PROBE_BURST", "1"))  # launches enqueued back to back between the two events
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        best, k = 1e9, 0
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(BURST):
                if L.offt_hipk_fft_pass(C.byref(d), src.data_ptr() + k * T * M2 * n * 16, dst.data_ptr() + k * (p2 * blk * 2 + 64) * 8, stream):
                    raise SystemExit("pass failed")
                k = (k + 1) % NT
            e1.record()
            torch.cuda.synchronize()
            if rep:
                best = min(best, e0.elapsed_time(e1) * 1e-3 / BURST)
        bytes_ = 2.0 * 16 * T * M2 * n
        print(f"K1 {p1}x{p2}: tile {T} x {M2} lines, row pitch {row * 16} B, plane pitch {plane * 16} B, peer blocks {blk * 16 / 2**20:.2f} MiB apart: "
              f"{best * 1e6:.1f} us = {bytes_ / best / 8e12 * 100:.1f} % of 8 TB/s  (tiles cycled {NT}, burst {BURST})", flush=True)
        del dst


if __name__ == "__main__":
    main()
