"""-m gpu: the kernel ABI (offt_hipk_fft_pass) against the test-only CPU interpreter of pass descriptors,
on randomised descriptors: every (in_contig, out_contig) flavour, per-peer splits on either side (power-of-two,
any other even length, uneven F/F+1), blocks addressed through per-block base tables (the blocks of one pass in
different places: self-bypass and direct-store exchange), both batch dimensions, ragged column panels, inverse,
scale, real input, f32.  This is what the fused pack/unpack of the multi-GPU schedules rests on, and a one-GPU
box cannot exercise those schedules with real peers."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from offt_amd import api

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Desc(C.Structure):
    _fields_ = [("n", C.c_int), ("precision", C.c_int), ("direction", C.c_int), ("ncols", C.c_int),
                ("nb1", C.c_int), ("nb2", C.c_int),
                ("in_axis_stride", C.c_longlong), ("in_col_stride", C.c_longlong), ("in_b1_stride", C.c_longlong),
                ("in_b2_stride", C.c_longlong),
                ("out_axis_stride", C.c_longlong), ("out_col_stride", C.c_longlong), ("out_b1_stride", C.c_longlong),
                ("out_b2_stride", C.c_longlong),
                ("in_split", C.c_int), ("in_split_nfloor", C.c_int), ("out_split", C.c_int), ("out_split_nfloor", C.c_int),
                ("in_block_stride", C.c_longlong), ("out_block_stride", C.c_longlong),
                ("in_block_tab", C.c_void_p), ("out_block_tab", C.c_void_p),
                ("in_contig", C.c_int), ("out_contig", C.c_int), ("variant", C.c_int), ("scale", C.c_double),
                ("real_input", C.c_int), ("out_keep", C.c_int), ("no_pairs", C.c_int), ("tw4", C.c_void_p), ("tw4_b1", C.c_int), ("tw4_n2", C.c_int)]


def layout(rng, n, ncols, nb1, nb2, split, nfloor, contig):
    """random non-overlapping strides for a [b2][b1][...] array whose inner two dims are (axis, col) in the order
    `contig` asks for; with a split the axis is cut into blocks that sit `block_stride` apart"""
    nblk = 1
    inner = n
    if split or nfloor:
        big = split + (1 if nfloor else 0)
        nblk = nfloor + (n - split * nfloor + big - 1) // big if nfloor else (n + split - 1) // split
        inner = big
    pad = int(rng.integers(0, 3))
    if contig:
        axis, col = 1, inner + pad
        plane = col * ncols
    else:
        col, axis = 1, ncols + pad
        plane = axis * inner
    blk = plane + int(rng.integers(0, 5))
    b1 = blk * nblk + int(rng.integers(0, 4))
    b2 = b1 * nb1 + int(rng.integers(0, 4))
    total = b2 * nb2 + 8
    return dict(axis=axis, col=col, b1=b1, b2=b2, blk=blk if (split or nfloor) else 0, total=total)


def make_case(rng, n, prec, big_grid=False):
    d = Desc()
    d.n, d.precision = n, prec
    d.direction = int(rng.choice([-1, -1, 1]))
    d.ncols, d.nb1, d.nb2 = int(rng.integers(1, 21)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
    if big_grid:  # hundreds to thousands of panels: the XCD-aware panel order and its unmapped tail
        d.ncols, d.nb1, d.nb2 = int(rng.integers(100, 700)), int(rng.integers(3, 12)), int(rng.integers(1, 4))
    d.in_contig, d.out_contig = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    d.variant = -1
    d.scale = float(rng.choice([1.0, 0.5, 1.0 / n]))
    d.real_input = 0
    d.out_keep = int(rng.integers(0, 2))  # cache-keeping stores where a twin kernel exists (contig-in / strided-out defaults)

    def pick_split():
        kind = rng.integers(0, 3)
        if kind == 0 or n < 4:
            return 0, 0
        if kind == 1:  # even split
            cands = [f for f in range(1, n) if n % f == 0]
            return int(rng.choice(cands)), 0
        p = int(rng.integers(2, min(n, 6) + 1))  # uneven: first p-b peers hold F, the rest F+1 (offt-compute.c:132-144)
        F, b = n // p, n % p
        return (F, p - b) if b else (F, 0)
    d.in_split, d.in_split_nfloor = pick_split()
    d.out_split, d.out_split_nfloor = pick_split()
    if rng.integers(0, 6) == 0 and d.direction < 0:  # real-input z pass
        d.real_input, d.in_contig, d.in_split, d.in_split_nfloor = 1, 1, 0, 0
    li = layout(rng, n, d.ncols, d.nb1, d.nb2, d.in_split, d.in_split_nfloor, d.in_contig)
    lo = layout(rng, n, d.ncols, d.nb1, d.nb2, d.out_split, d.out_split_nfloor, d.out_contig)
    if d.real_input:  # rows of n/2+1 complex slots hold n reals
        li = layout(rng, n // 2 + 1, d.ncols, d.nb1, d.nb2, 0, 0, 1)
    d.in_axis_stride, d.in_col_stride, d.in_b1_stride, d.in_b2_stride, d.in_block_stride = li["axis"], li["col"], li["b1"], li["b2"], li["blk"]
    d.out_axis_stride, d.out_col_stride, d.out_b1_stride, d.out_b2_stride, d.out_block_stride = lo["axis"], lo["col"], lo["b1"], lo["b2"], lo["blk"]
    return d, li["total"], lo["total"]


def nblocks(n, split, nfloor):
    if not (split or nfloor):
        return 0
    big = split + (1 if nfloor else 0)
    return nfloor + (n - split * nfloor + big - 1) // big if nfloor else (n + split - 1) // split


def with_tables(rng, d, nin, nout):
    """move the blocks of the split sides to shuffled places behind the array (per-block base tables): returns the
    enlarged element counts and the two tables (None where the side has no split).  The descriptor's block stride is
    set to a value that would be wrong, so a kernel that ignores the table cannot pass."""
    tabs = []
    sizes = [nin, nout]
    for side, (split, nfloor, blk) in enumerate(((d.in_split, d.in_split_nfloor, d.in_block_stride),
                                                 (d.out_split, d.out_split_nfloor, d.out_block_stride))):
        nb = nblocks(d.n, split, nfloor)
        if not nb or rng.integers(0, 3) == 0:
            tabs.append(None)
            continue
        span = sizes[side] + (sizes[side] & 1)               # (even: a column pair's 16 bytes stay aligned)
        slots = rng.permutation(nb + 2)[:nb]                 # every block gets a region of its own ...
        t = np.array([int(b) * blk + (int(s) + 1) * span for b, s in zip(range(nb), slots)], dtype=np.int64)
        keep = rng.integers(0, nb)                           # ... and one of them stays where the stride puts it
        t[keep] = int(keep) * blk
        sizes[side] = span * (nb + 4)
        tabs.append(t)
        if side == 0:
            d.in_block_stride = 1
        else:
            d.out_block_stride = 1
    return sizes[0], sizes[1], tabs


def run_both(L, CB, d, src, nout, ct, ft, tabs):
    """descriptor on the CPU interpreter (host tables) and on the GPU (device tables) -> (want, got)"""
    want = np.full(nout, 7 - 3j, dtype=ct)  # sentinel: untouched elements must stay untouched
    d.in_block_tab = tabs[0].ctypes.data if tabs[0] is not None else None
    d.out_block_tab = tabs[1].ctypes.data if tabs[1] is not None else None
    assert CB.cpu_backend_run_pass(C.byref(d), src.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p)) == 0
    dt = [torch.from_numpy(t.copy()).cuda() if t is not None else None for t in tabs]
    d.in_block_tab = dt[0].data_ptr() if dt[0] is not None else None
    d.out_block_tab = dt[1].data_ptr() if dt[1] is not None else None
    din = torch.from_numpy(src.view(ft).copy()).cuda()
    dout = torch.from_numpy(np.full(nout, 7 - 3j, dtype=ct).view(ft).copy()).cuda()
    torch.cuda.synchronize()
    rc = L.offt_hipk_fft_pass(C.byref(d), din.data_ptr(), dout.data_ptr(), None)
    assert rc == 0, L.offt_hipk_last_error()
    torch.cuda.synchronize()
    return want, dout.cpu().numpy().view(ct)


@pytest.fixture(scope="module")
def libs(built):
    L = api.lib()
    L.offt_hipk_fft_pass.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.offt_hipk_prepare.argtypes = [C.c_int, C.c_int]
    L.offt_hipk_last_error.restype = C.c_char_p
    CB = C.CDLL(os.path.join(ROOT, "tests", "libcpubackend.so"))
    CB.cpu_backend_run_pass.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_void_p]
    return L, CB


REPS = int(os.environ.get("OFFT_TEST_DESC_REPS", "6"))


@pytest.mark.parametrize("n", [2, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 6, 12, 15, 18, 30, 45, 90, 100, 127, 384,
                               768, 1000, 448, 896, 1001, 37, 254, 509, 1016, 1021, 2039])  # the last six: Bluestein
def test_random_descriptors(libs, n):
    L, CB = libs
    rng = np.random.default_rng(1000 + n)
    for prec in (api.F64, api.F32):
        assert L.offt_hipk_prepare(n, prec) == 0
        ft, ct = (np.float64, np.complex128) if prec == api.F64 else (np.float32, np.complex64)
        for it in range(REPS if n <= 512 else max(3, REPS // 2)):
            d, nin, nout = make_case(rng, n, prec)
            tabs = [None, None]
            if it % 2 == 1:  # every other case: blocks through per-block base tables
                nin, nout, tabs = with_tables(rng, d, nin, nout)
            src = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(ct)
            want, got = run_both(L, CB, d, src, nout, ct, ft, tabs)
            tol = 1e-13 if prec == api.F64 else 5e-6
            scale = np.abs(want).max()
            desc = {f: getattr(d, f) for f, _ in Desc._fields_}
            assert np.abs(got - want).max() <= tol * scale * max(1.0, np.log2(n)), desc


def make_pair_case(rng, n, variant):
    """a single-precision descriptor the column-pair kernels accept: even column count; a strided side with unit column
    stride and even other strides (the pair is 16 aligned bytes); a contiguous side with any strides; power-of-two
    per-peer blocks or none"""
    d = Desc()
    d.n, d.precision = n, api.F32
    d.direction = int(rng.choice([-1, -1, 1]))
    d.ncols, d.nb1, d.nb2 = 2 * int(rng.integers(1, 30)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
    d.in_contig, d.out_contig = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    d.variant, d.scale, d.real_input = variant, float(rng.choice([1.0, 0.5, 1.0 / n])), 0
    d.out_keep = int(rng.integers(0, 2))

    def lay(contig):
        split = int(rng.choice([0, 0, 2, 8, n // 4, n // 2]))
        nblk, inner = (n // split, split) if split else (1, n)
        if contig:
            axis, col = 1, inner + int(rng.integers(0, 4))
            plane = col * d.ncols
            ev = 1
        else:
            col, axis = 1, d.ncols + 2 * int(rng.integers(0, 3))
            plane = axis * inner
            ev = 2
        blk = plane + ev * int(rng.integers(0, 4))
        b1 = blk * nblk + ev * int(rng.integers(0, 3))
        b2 = b1 * d.nb1 + ev * int(rng.integers(0, 3))
        return split, axis, col, b1, b2, (blk if split else 0), b2 * d.nb2 + 8
    d.in_split, d.in_axis_stride, d.in_col_stride, d.in_b1_stride, d.in_b2_stride, d.in_block_stride, nin = lay(d.in_contig)
    d.out_split, d.out_axis_stride, d.out_col_stride, d.out_b1_stride, d.out_b2_stride, d.out_block_stride, nout = lay(d.out_contig)
    d.in_split_nfloor = d.out_split_nfloor = 0
    return d, nin, nout


@pytest.mark.parametrize("n", [512, 1024, 2048, 4096])
def test_column_pair_kernels(libs, n):
    """the single-precision column-pair kernels (T = f32x2: two adjacent columns per lane, 16 B per lane on a strided
    side), every registered pair variant forced through descriptor variant 200 + id, against the CPU interpreter; plus
    descriptors they must NOT take (odd column count, odd strides on a strided side, a base pointer off the 16-B grid),
    which have to come out right on the one-column kernels"""
    L, CB = libs
    L.offt_hipk_kernel_name.restype = C.c_char_p
    L.offt_hipk_kernel_name.argtypes = [C.POINTER(Desc)]
    rng = np.random.default_rng(4200 + n)
    assert L.offt_hipk_prepare(n, api.F32) == 0
    ct = np.complex64

    def check(d, nin, nout, shift=0, tables=False):
        tabs = [None, None]
        if tables:
            nin, nout, tabs = with_tables(rng, d, nin, nout)
        src = (rng.standard_normal(nin + 1) + 1j * rng.standard_normal(nin + 1)).astype(ct)
        want = np.full(nout + 1, 7 - 3j, dtype=ct)
        d.in_block_tab = tabs[0].ctypes.data if tabs[0] is not None else None
        d.out_block_tab = tabs[1].ctypes.data if tabs[1] is not None else None
        assert CB.cpu_backend_run_pass(C.byref(d), src[shift:].ctypes.data_as(C.c_void_p), want[shift:].ctypes.data_as(C.c_void_p)) == 0
        dt = [torch.from_numpy(t.copy()).cuda() if t is not None else None for t in tabs]
        d.in_block_tab = dt[0].data_ptr() if dt[0] is not None else None
        d.out_block_tab = dt[1].data_ptr() if dt[1] is not None else None
        din = torch.from_numpy(src.view(np.float32).copy()).cuda()
        dout = torch.from_numpy(np.full(nout + 1, 7 - 3j, dtype=ct).view(np.float32).copy()).cuda()
        torch.cuda.synchronize()
        rc = L.offt_hipk_fft_pass(C.byref(d), din.data_ptr() + 8 * shift, dout.data_ptr() + 8 * shift, None)
        assert rc == 0, L.offt_hipk_last_error()
        torch.cuda.synchronize()
        got = dout.cpu().numpy().view(ct)
        desc = {f: getattr(d, f) for f, _ in Desc._fields_}
        assert np.abs(got - want).max() <= 5e-6 * np.abs(want).max() * np.log2(n), desc

    seen = 0
    for vid in range(4):
        for _ in range(6):
            d, nin, nout = make_pair_case(rng, n, 200 + vid)
            name = L.offt_hipk_kernel_name(C.byref(d)).decode()
            if name != "fft_panel_k<pairs>":
                assert vid > 0, "no column-pair kernel registered for n=%d" % n
                break
            seen += 1
            check(d, nin, nout, tables=bool(seen % 2))
    assert seen >= 6
    # not eligible: odd column count / an odd stride on a strided side / a misaligned base -> one-column kernels
    for _ in range(6):
        d, nin, nout = make_pair_case(rng, n, 200)
        kind = int(rng.integers(0, 3))
        if kind == 0:
            d.ncols -= 1
        elif kind == 1 and not d.in_contig:
            d.in_b1_stride += 1
            d.in_b2_stride += d.nb1
            nin += d.nb1 * d.nb2 + 8
        elif kind == 1 and not d.out_contig:
            d.out_b1_stride += 1
            d.out_b2_stride += d.nb1
            nout += d.nb1 * d.nb2 + 8
        if kind < 2 and (kind == 0 or not (d.in_contig and d.out_contig)):
            assert L.offt_hipk_kernel_name(C.byref(d)).decode() == "fft_panel_k"
        check(d, nin, nout, shift=1 if kind == 2 else 0)


@pytest.mark.parametrize("n", [16, 64, 96, 100, 127, 256])
def test_many_panels_xcd_order(libs, n):
    """grids of 300 .. 8000 panels with ragged last panels: the XCD-aware panel renumbering (panel_of_block) is a
    bijection and its tail (blocks that do not fill 8 x 32 panels) keeps launch order"""
    L, CB = libs
    rng = np.random.default_rng(7000 + n)
    for prec in (api.F64, api.F32):
        assert L.offt_hipk_prepare(n, prec) == 0
        ft, ct = (np.float64, np.complex128) if prec == api.F64 else (np.float32, np.complex64)
        for _ in range(3):
            d, nin, nout = make_case(rng, n, prec, big_grid=True)
            src = (rng.standard_normal(nin) + 1j * rng.standard_normal(nin)).astype(ct)
            want = np.full(nout, 7 - 3j, dtype=ct)
            assert CB.cpu_backend_run_pass(C.byref(d), src.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p)) == 0
            din = torch.from_numpy(src.view(ft).copy()).cuda()
            dout = torch.from_numpy(np.full(nout, 7 - 3j, dtype=ct).view(ft).copy()).cuda()
            torch.cuda.synchronize()
            rc = L.offt_hipk_fft_pass(C.byref(d), din.data_ptr(), dout.data_ptr(), None)
            assert rc == 0, L.offt_hipk_last_error()
            torch.cuda.synchronize()
            got = dout.cpu().numpy().view(ct)
            tol = 1e-13 if prec == api.F64 else 5e-6
            desc = {f: getattr(d, f) for f, _ in Desc._fields_}
            assert np.abs(got - want).max() <= tol * np.abs(want).max() * max(1.0, np.log2(n)), desc


def test_split_routing(libs):
    """which kernel a descriptor resolves to: power-of-two blocks -> fft_panel_k; blocks of any other length and
    the reference's uneven F / F+1 blocks -> the length's fft_panelx_k instance; a length without a register
    kernel -> the any-length kernel"""
    L, _ = libs
    L.offt_hipk_kernel_name.restype = C.c_char_p
    L.offt_hipk_kernel_name.argtypes = [C.POINTER(Desc)]

    def name(n, in_split=0, in_nfloor=0, out_split=0, out_nfloor=0, prec=api.F64):
        d = Desc()
        d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2 = n, prec, -1, 8, 1, 1
        d.in_contig = d.out_contig = 1
        d.variant, d.scale = -1, 1.0
        d.in_split, d.in_split_nfloor, d.out_split, d.out_split_nfloor = in_split, in_nfloor, out_split, out_nfloor
        return L.offt_hipk_kernel_name(C.byref(d)).decode()
    assert name(1024) == "fft_panel_k"
    assert name(1024, out_split=128) == "fft_panel_k"
    assert name(1024, out_split=341, out_nfloor=2) == "fft_panelx_k"      # 1024 over 3 peers: 341, 341, 342
    assert name(1024, in_split=170, in_nfloor=2) == "fft_panelx_k"        # 1024 over 6 peers
    assert name(768, out_split=96) == "fft_panelx_k"
    assert name(1000, in_split=142, in_nfloor=1) == "fft_panelx_k"        # 1000 over 7 peers
    assert L.offt_hipk_prepare(127, api.F64) == 0 and L.offt_hipk_prepare(127, api.F32) == 0
    assert name(127) == "fft_bluestein_k"    # a prime: chirp-z on the 256-point panel machinery (round 2)
    assert name(4099) == "fft_mixed_k"       # 2 n - 1 > 4096: the any-length kernel
    assert name(1024, out_split=341, out_nfloor=2, prec=api.F32) == "fft_panelx_k"  # f32 any-split instances (round 2)
    assert name(2048, in_split=292, in_nfloor=3, prec=api.F32) == "fft_panelx_k"    # 2048 over 7 peers, single precision
    assert name(127, prec=api.F32) == "fft_bluestein_k"

