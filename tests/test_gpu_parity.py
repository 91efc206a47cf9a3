"""-m gpu: the HIP path, called through the C ABI, against the oracle (CPU restatement of the
reference), the committed golden fixtures and size-independent properties at full size.

Tolerances (BASELINE.md 4 / SURVEY.md 8d): double rel-L2 <= 1e-13 and max-abs/max <= 1e-12;
single rel-L2 <= 5e-6."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_lib as O
from gpu_util import gpu_fft, make_input, read_output, rel, gpu_roundtrip
from offt_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL64, TOLMAX64, TOL32 = 1e-13, 1e-12, 5e-6


def check64(got, want):
    assert rel(got, want) <= TOL64
    assert np.abs(got - want).max() / np.abs(want).max() <= TOLMAX64


LAYOUTS = [dict(S=1), dict(), dict(eq=1)]


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_cube_vs_oracle(built, n, layout):
    eq = layout.get("eq", 0)
    params = {k: v for k, v in layout.items() if k != "eq"}
    got, c = gpu_fft((n, n, n), is_equalxy=eq, **params)
    want, comms, _ = O.world_fft(n, n, n, 1, kind=1, is_equalxy=eq, **params)
    check64(got, want)
    oc = comms[0]
    for k in ("istart", "isize", "istride", "ostart", "osize", "ostride", "M1", "M2", "M3", "M4"):
        assert c[k] == oc[k], k


@pytest.mark.parametrize("n", [8, 16, 20])
def test_numpy_golden_fixtures(built, n):
    F = np.load(os.path.join(G, f"numpy_fftn_hash_{n}.npz"))["F"]
    for layout in LAYOUTS:
        eq = layout.get("eq", 0)
        got, _ = gpu_fft((n, n, n), is_equalxy=eq, **{k: v for k, v in layout.items() if k != "eq"})
        check64(got, F)


def test_reference_dump_fixture(built):
    """the compiled reference's own full-grid output (18^3, survey run) -- any decomposition gives the same X[k]"""
    d = np.load(os.path.join(G, "ref_n18_p6_p1-2_S1.npz"))
    ref = np.zeros((18, 18, 18), dtype=complex)
    r = d["rank_xyz"].astype(int)
    ref[r[:, 1], r[:, 2], r[:, 3]] = d["value"]
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft((18, 18, 18), **layout)
        check64(got, ref)


@pytest.mark.parametrize("shape", [(64, 32, 16), (16, 128, 64), (256, 8, 4), (4, 512, 8), (8, 4, 1024), (2048, 4, 4),
                                   (4, 4, 4096), (2, 2, 2), (4, 2, 8)])
def test_non_cubic_pow2(built, shape):
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft(shape, **layout)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **layout)
        check64(got, want)


@pytest.mark.parametrize("shape", [(20, 20, 20), (18, 12, 30), (7, 5, 3), (1, 9, 1), (1, 1, 1), (100, 3, 6), (3, 96, 50),
                                   (120, 96, 100), (127, 6, 4), (4, 1000, 8), (2, 2, 4093), (384, 4, 4), (6, 5000, 2)])
def test_any_length_generic_kernel(built, shape):
    """lengths without a register fast path (FFTW accepts any N): mixed-radix any-length kernel (composite,
    prime, large prime 4093, > 4096), same tolerance"""
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft(shape, **layout)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **layout)
        check64(got, want)


@pytest.mark.parametrize("shape", [(96, 96, 96), (120, 100, 144), (768, 6, 10), (6, 768, 10), (10, 6, 768), (1000, 4, 12), (4, 12, 1000),
                                   (12, 1536, 4), (640, 8, 384), (8, 1200, 6), (2, 6, 3072), (3200, 2, 6), (1920, 4, 2),
                                   (896, 4, 6), (6, 448, 10), (1001, 2, 4), (4, 6, 1344), (224, 224, 56)])
def test_mixed_radix_register_kernel(built, shape):
    """lengths 2^a 3^b 5^c (and 7-smooth ones, 1001 = 7 * 11 * 13) with a compile-time mixed-radix panel kernel
    (fft_panelx_k), every layout flavour"""
    L = api.lib()
    assert any(L.offt_hipk_has_fast_path(n, api.F64) for n in shape if n & (n - 1)), shape
    for layout in (dict(S=1), dict(), dict(eq=1)):
        eq = layout.get("eq", 0)
        if eq and shape[0] != shape[1]:
            continue
        params = {k: v for k, v in layout.items() if k != "eq"}
        got, _ = gpu_fft(shape, is_equalxy=eq, **params)
        want, _, _ = O.world_fft(*shape, 1, kind=1, is_equalxy=eq, **params)
        check64(got, want)
    # inverse(forward(x)) / N == x, and the r2c z pass on the mixed-radix length
    f = O.hash_field(*shape)
    po = api.offt_3d_init(*shape)
    try:
        dev, idx = make_input(po, f)
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
        back = dev.cpu().numpy().view(np.complex128)[idx].reshape(shape) / np.prod(shape)
        assert rel(back, f) <= TOL64
    finally:
        api.offt_3d_fin(po)
    if shape[2] % 2 == 0:
        got, _ = gpu_fft(shape, is_r2c=1)
        want, _, _ = O.world_fft(*shape, 1, kind=1, is_r2c=1)
        check64(got, want)


# (every shape compiles a kernel at plan time, 10-15 s each: 1296, 2500 and 608 = 19*32 were dropped from the list to keep the tier short)
@pytest.mark.parametrize("shape", [(432, 6, 10), (1728, 4, 2), (324, 324, 8), (6, 2187, 2), (4, 4, 3528),
                                   (1088, 4, 2), (4, 4, 992), (2, 2, 1334)])  # 17*64, 31*32, 2*23*29
def test_plan_time_specialised_kernel(built, shape):
    """a 31-smooth length of 256 .. 4096 points without a precompiled panel kernel: offt_3d_init has hipRTC compile
    fft_panelx_k for it (offt_hipk_prepare); same tolerances, every layout, f32 too"""
    L = api.lib()
    big = max(shape)
    for prec, tol in ((api.F64, None), (api.F32, TOL32)):
        for layout in (dict(S=1), dict()):
            got, _ = gpu_fft(shape, precision=prec, **layout)
            assert L.offt_hipk_has_fast_path(big, prec) == 1, "no plan-time kernel was built (hipRTC missing?)"
            assert b"plan-time" in L.offt_hipk_variant_name(big, prec, -1)
            want, _, _ = O.world_fft(*shape, 1, kind=1, **layout)
            if tol is None:
                check64(got, want)
            else:
                assert rel(got.astype(np.complex128), want) < tol
    if shape[2] % 2 == 0:  # the real-input z pass has plan-time instances too
        got, _ = gpu_fft(shape, is_r2c=1)
        want, _, _ = O.world_fft(*shape, 1, kind=1, is_r2c=1)
        check64(got, want)


@pytest.mark.parametrize("shape", [(127, 8, 4), (4, 254, 6), (6, 4, 1016), (1021, 4, 2), (2, 2, 2039), (37, 74, 41), (508, 2, 127)])
def test_bluestein_lengths(built, shape):
    """lengths with a prime factor > 31 (FFTW takes any N): Bluestein's chirp-z on the power-of-two panel machinery,
    one HBM round trip (offt_bluestein.hpp) -- every layout, inverse round trip, single precision"""
    L = api.lib()
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft(shape, **layout)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **layout)
        check64(got, want)
    got, _ = gpu_fft(shape, precision=api.F32)
    want, _, _ = O.world_fft(*shape, 1, kind=1)
    assert rel(got.astype(np.complex128), want) < TOL32
    if shape[2] == 1016:  # real input along a Bluestein length: gathered into scratch lines, the same kernel, n/2 + 1 outputs kept
        for sh in (shape, (4, 4, 2038)):
            got, _ = gpu_fft(sh, is_r2c=1)
            assert rel(got, np.fft.rfftn(O.hash_field(*sh).real)) < TOL64, sh
    # forward then inverse gives the input back times E
    po = api.offt_3d_init(*shape)
    f = O.hash_field(*shape)
    dev, idx = make_input(po, f)
    x0 = dev.clone()
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
    back = dev.cpu().numpy().view(np.complex128)[idx].reshape(shape) / np.prod(shape)
    api.offt_3d_fin(po)
    assert rel(back, f) < TOL64
    # the pass really ran on the Bluestein kernel
    import ctypes as C
    from test_gpu_descriptors import Desc
    L.offt_hipk_kernel_name.restype = C.c_char_p
    L.offt_hipk_kernel_name.argtypes = [C.POINTER(Desc)]
    big = max(s for s in shape if any(s % p == 0 for p in range(37, s + 1) if all(p % q for q in range(2, int(p ** 0.5) + 1))))
    d = Desc()
    d.n, d.precision, d.direction, d.ncols, d.nb1, d.nb2, d.in_contig, d.out_contig, d.variant, d.scale = big, api.F64, -1, 8, 1, 1, 1, 1, -1, 1.0
    assert L.offt_hipk_kernel_name(C.byref(d)) == b"fft_bluestein_k", big


def test_mixed_radix_full_size_768_properties(built):
    """768^3 (3 * 2^8) on the mixed-radix panel kernel: Parseval, DC term, forward/inverse round trip"""
    n = 768
    L = api.lib()
    po = api.offt_3d_init(n, n, n)
    try:
        nel = api.local_elems(po)
        dev = torch.zeros(nel * 2, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        L.offt_hip_fill_input(po, dev.data_ptr(), 1)
        torch.cuda.synchronize()
        x = dev.clone()
        e_in = float((x * x).sum())
        dc = torch.view_as_complex(x.view(-1, 2)).sum()
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        torch.cuda.synchronize()
        X = torch.view_as_complex(dev.view(-1, 2))
        e_out = float((dev * dev).sum())
        assert abs(e_out / (e_in * n ** 3) - 1) < 1e-12
        assert abs(complex(X[0]) - complex(dc)) <= 1e-9 * abs(complex(dc)) + 1e-6
        api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
        torch.cuda.synchronize()
        err = float(((dev / n ** 3 - x) ** 2).sum()) ** 0.5 / e_in ** 0.5
        assert err < TOL64
    finally:
        api.offt_3d_fin(po)


def test_ramp_closed_form_spot_values(built):
    """run-fft -v prints out[0,0,0..3]; for the ramp these have a closed form"""
    rec = json.load(open(os.path.join(G, "survey_recorded.json")))
    for n in (128, 256):
        got, _ = gpu_fft((n, n, n), field=O.ramp_field(n, n, n))
        assert got[0, 0, 0].real == n ** 3 * 111 * (n - 1) / 2 and got[0, 0, 0].imag == 0
        k = np.arange(1, 4)
        cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n))
        assert np.abs(got[0, 0, 1:4] - cf).max() / np.abs(cf).max() < 1e-13
        if n == 128:
            assert got[0, 0, 0].real == rec["ramp_128_p2_X000"]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_forward_inverse_roundtrip_512(built, layout):
    """BASELINE config[1]: 512^3 double-complex forward + inverse on one MI355X"""
    n = 512
    eq = layout.get("eq", 0)
    po = api.offt_3d_init(n, n, n, custom_params=api.make_params(**{k: v for k, v in layout.items() if k != "eq"}),
                          is_equalxy=eq)
    L = api.lib()
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()  # the plan runs on its own non-blocking stream: order it after torch's fill
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    x0 = dev.clone()
    torch.cuda.synchronize()
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    # Parseval on the forward result
    e_in, e_out = float((x0 * x0).sum()), float((dev * dev).sum())
    assert abs(e_out / n ** 3 - e_in) / e_in < 1e-13
    torch.cuda.synchronize()
    api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
    dev /= n ** 3
    err = float(torch.linalg.vector_norm(dev - x0) / torch.linalg.vector_norm(x0))
    api.offt_3d_fin(po)
    assert err < TOL64


def test_full_size_1024_properties(built):
    """1024^3 double-complex (BASELINE config[2]): closed-form ramp spots + Parseval + linearity probe"""
    n = 1024
    po = api.offt_3d_init(n, n, n)
    L = api.lib()
    c = api.comm_dict(po)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 0)  # harness ramp
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    cv = torch.view_as_complex(dev.view(-1, 2))
    os0, os1, os2 = c["ostride"]
    assert complex(cv[0]) == n ** 3 * 111 * (n - 1) / 2
    for k in (1, 2, 3, 511):
        cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n))
        assert abs(complex(cv[k * os2]) - cf) / abs(cf) < 1e-12   # X[0,0,k]
        assert abs(complex(cv[k * os1]) - 10 * cf) / abs(10 * cf) < 1e-12  # X[0,k,0] = 10 * ...
        assert abs(complex(cv[k * os0]) - 100 * cf) / abs(100 * cf) < 1e-12
    # everything off the three axes is zero for a separable ramp
    assert abs(complex(cv[os0 + os1 + os2])) / n ** 3 < 1e-9
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    e_in = float((dev * dev).sum())
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    e_out = float((dev * dev).sum())
    api.offt_3d_fin(po)
    assert abs(e_out / n ** 3 - e_in) / e_in < 1e-13


@pytest.mark.parametrize("n", [16, 64, 256])
def test_single_precision(built, n):
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft((n, n, n), precision=api.F32, **layout)
        want = np.fft.fftn(O.hash_field(n, n, n).astype(np.complex64).astype(np.complex128))
        assert rel(got.astype(np.complex128), want) <= TOL32


@pytest.mark.parametrize("shape", [(8192, 4, 8), (4, 8192, 8), (8, 4, 8192)])
def test_long_lines_8192(built, shape):
    """8192-point lines (the longest register kernel; one column per workgroup in f64): every axis, both precisions"""
    for layout in (dict(S=1), dict()):
        got, _ = gpu_fft(shape, **layout)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **layout)
        check64(got, want)
    got, _ = gpu_fft(shape, precision=api.F32)
    want, _, _ = O.world_fft(*shape, 1, kind=1)
    assert rel(got.astype(np.complex128), want) <= TOL32


@pytest.mark.parametrize("shape", [(2048, 8, 8), (8, 2048, 8), (8, 8, 2048)])
def test_single_precision_2048_sides(built, shape):
    """BASELINE configs[4] kernels: a 2048-point single-precision pass on every axis, every output layout, against
    the oracle (double) with the single-precision tolerance"""
    for layout in (dict(S=1), dict(), dict(is_equalxy=1)):
        if layout.get("is_equalxy") and shape[0] != shape[1]:
            continue
        got, _ = gpu_fft(shape, precision=api.F32, **layout)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **{("is_equalxy" if k == "is_equalxy" else k): v for k, v in layout.items()})
        assert rel(got.astype(np.complex128), want) <= TOL32, (shape, layout)
    # forced tile pipeline (the multi-rank kernels' addressing) in single precision
    os.environ["OFFT_FORCE_PIPELINE"] = "1"
    try:
        got, _ = gpu_fft(shape, precision=api.F32, T1=4, T2=4)
        want, _, _ = O.world_fft(*shape, 1, kind=1, T1=4, T2=4)
        assert rel(got.astype(np.complex128), want) <= TOL32, shape
    finally:
        del os.environ["OFFT_FORCE_PIPELINE"]


def test_full_size_2048_single_precision_properties(built):
    """2048^3 single-complex on one GPU (64 GiB grid + 64 GiB scratch; BASELINE configs[4]'s grid): closed-form ramp
    spots and Parseval, computed in place with chunked reductions (no full clones)"""
    n = 2048
    po = api.offt_3d_init(n, n, n, precision=api.F32)
    L = api.lib()
    c = api.comm_dict(po)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def energy():  # sum of squares in float64, 2^28 floats at a time
        tot = 0.0
        step = 1 << 28
        for i in range(0, dev.numel(), step):
            tot += float(dev[i:i + step].double().square().sum())
        return tot

    # ramp scaled to keep single precision honest: re = (z + 10 y + 100 x) / 2^18, so that X[0,0,0] ~ 2^33 * 0.43
    L.offt_hip_fill_input(po, dev.data_ptr(), 0)
    torch.cuda.synchronize()
    L.offt_hip_set_output_scale(po, 2.0 ** -33)
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    cv = torch.view_as_complex(dev.view(-1, 2))
    os0, os1, os2 = c["ostride"]
    x000 = n ** 3 * 111 * (n - 1) / 2 * 2.0 ** -33
    assert abs(complex(cv[0]) - x000) / x000 < 2e-6
    cf1 = abs(n ** 3 * (-0.5 + 0.5j / np.tan(np.pi / n)) * 2.0 ** -33)
    for k in (1, 2, 3, 1023):
        cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n)) * 2.0 ** -33
        # single precision: errors scale with the large low-frequency bins of the ramp, not with the bin itself
        # (|X[0,0,1023]| is 650 times smaller than |X[0,0,1]|): relative to the bin for k <= 3, to |X[0,0,1]| beyond
        ref = abs(cf) if k <= 3 else cf1
        assert abs(complex(cv[k * os2]) - cf) / ref < 5e-6
        assert abs(complex(cv[k * os1]) - 10 * cf) / (10 * ref) < 5e-6
        assert abs(complex(cv[k * os0]) - 100 * cf) / (100 * ref) < 5e-6
    assert abs(complex(cv[os0 + os1 + os2])) / abs(x000) < 1e-6
    # Parseval on the seeded hash field: sum |X|^2 = E * sum |x|^2 (unnormalised forward transform)
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    torch.cuda.synchronize()
    e_in = energy()
    L.offt_hip_set_output_scale(po, 2.0 ** -16)   # exact power of two: |X|^2 scaled by 2^-32 = 1 / sqrt(E)^... see below
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    e_out = energy()
    # E = 2^33, scale^2 = 2^-32: e_out = 2^-32 * 2^33 * e_in = 2 e_in
    assert abs(e_out / (2.0 * e_in) - 1.0) < 1e-5
    # FULL-GRID rel-L2 at this size: forward, then the inverse, against the input (scales 2^-16 and 2^-17: together 1 / E,
    # exact powers of two) -- every element of the 2^33-point grid takes part, chunked reductions in float64
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    torch.cuda.synchronize()
    x0 = dev.clone()
    torch.cuda.synchronize()
    L.offt_hip_set_output_scale(po, 2.0 ** -16)
    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    L.offt_hip_set_output_scale(po, 2.0 ** -17)
    api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
    num = den = 0.0
    step = 1 << 28
    for i in range(0, dev.numel(), step):
        a, b = dev[i:i + step].double(), x0[i:i + step].double()
        num += float((a - b).square().sum())
        den += float(b.square().sum())
    api.offt_3d_fin(po)
    assert (num / den) ** 0.5 < TOL32, (num / den) ** 0.5


def test_full_size_1024_single_precision_full_grid_vs_double(built):
    """1024^3: the single-precision transform against the double-precision one of the same field, FULL grid, rel-L2 <= 5e-6
    (the double-precision result at this size is itself checked by closed forms and Parseval above, and against the
    oracle at the sizes the oracle finishes)"""
    n = 1024
    L = api.lib()
    po64 = api.offt_3d_init(n, n, n)
    d64 = torch.zeros(api.local_elems(po64) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po64, d64.data_ptr(), 1)
    api.offt_3d_execute(po64, d64.data_ptr(), d64.data_ptr())
    api.offt_3d_fin(po64)
    po32 = api.offt_3d_init(n, n, n, precision=api.F32)
    d32 = torch.zeros(api.local_elems(po32) * 2, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po32, d32.data_ptr(), 1)
    api.offt_3d_execute(po32, d32.data_ptr(), d32.data_ptr())
    api.offt_3d_fin(po32)
    num = den = 0.0
    step = 1 << 27
    for i in range(0, d64.numel(), step):
        a, b = d32[i:i + step].double(), d64[i:i + step]
        num += float((a - b).square().sum())
        den += float(b.square().sum())
    assert (num / den) ** 0.5 < TOL32, (num / den) ** 0.5


@pytest.mark.parametrize("shape,prec,kw", [((8, 8, 6000), api.F64, {}), ((16384, 4, 4), api.F64, {}), ((4, 12000, 4), api.F32, {}),
                                           ((8, 6000, 8), api.F64, dict(S=1)), ((8, 8, 16384), api.F32, dict(S=1)),
                                           ((8192, 8, 8), api.F64, {}), ((8, 8192, 8), api.F64, dict(S=1)), ((8, 8, 8192), api.F32, {}),
                                           # lengths the any-length kernel could take, routed here because they have a fused split
                                           ((8, 8, 5000), api.F64, {}), ((4, 10000, 4), api.F32, {}), ((4800, 4, 4), api.F64, dict(S=1)),
                                           ((8, 8, 4076), api.F64, {}), ((4, 4076, 4), api.F32, dict(S=1))])  # 4 x 1019: the long factor on the Bluestein panel kernel
def test_long_lines_four_step(built, shape, prec, kw):
    """lines no single kernel takes (above 5120 double / 10240 single points; FFTW plans any N, offt-compute.c:335-341,
    416-425) run as a four-step decomposition n = n1 n2 -- two sub-passes of the library's own kernels and a twiddle sweep
    -- on every axis and in every pass flavour; 8192 has a one-column register kernel whose strided flavours go the same
    way.  Forward against the oracle, inverse round trip."""
    got, c = gpu_fft(shape, precision=prec, **kw)
    want, _, _ = O.world_fft(*shape, 1, kind=1, **kw)
    assert rel(got, want) < (TOL64 if prec == api.F64 else TOL32), (shape, kw)
    back = gpu_roundtrip(shape, precision=prec, **kw)
    assert back < (TOL64 if prec == api.F64 else TOL32)


@pytest.mark.parametrize("shape,prec,r2c,kw", [((8, 8, 10007), api.F64, 0, {}), ((4, 20011, 4), api.F32, 0, {}), ((10007, 4, 4), api.F64, 0, dict(S=1)),
                                               ((8, 8, 3057), api.F64, 0, {}), ((4, 3057, 4), api.F32, 0, dict(S=1)),  # 3 x 1019: the any-length kernel's radix would be 1019
                                               ((4, 4, 3057 * 2), api.F64, 1, {}),
                                               # 2 x 3061 (prime): a four-step line whose long factor goes through scratch lines itself -- and, with real
                                               # input, is gathered into scratch lines first: three levels of scratch, each with its own buffers
                                               ((4, 4, 6122), api.F64, 1, {}), ((4, 6122, 4), api.F64, 0, {}),
                                               ((8, 8, 6000), api.F64, 1, {}), ((4, 4, 16384), api.F32, 1, {}), ((4, 4, 10007), api.F64, 1, {})])
def test_long_lines_through_scratch(built, shape, prec, r2c, kw):
    """what the four-step decomposition leaves: a long PRIME line (no n1 n2 to decompose along) runs as a Bluestein
    convolution on power-of-two lines of >= 2n - 1 points, a long real-input line as a complex line whose first n/2 + 1
    outputs are kept -- both through dense scratch lines with the library's own long-line path.  Against numpy (the oracle's
    prime lines are plain O(n^2) sums); complex: inverse round trip as well."""
    f = O.hash_field(*shape)
    got, _ = gpu_fft(shape, field=f, precision=prec, is_r2c=r2c, **kw)
    want = np.fft.rfftn(f.real) if r2c else np.fft.fftn(f)
    assert rel(got, want) < (TOL64 if prec == api.F64 else TOL32), (shape, kw)
    if not r2c:
        assert gpu_roundtrip(shape, precision=prec, **kw) < (TOL64 if prec == api.F64 else TOL32)


def test_lengths_without_any_kernel_are_refused_at_plan_time(built):
    """with the Bluestein net switched off (OFFT_BLUESTEIN_LONG=0) a long prime line has no kernel: that must fail in
    offt_3d_init, not on every execute (the switch is read once per process, hence a process of its own)"""
    code = ("import sys; sys.path[:0] = [%r, %r]\nfrom offt_amd import api\n"
            "try:\n    api.offt_3d_init(8, 8, 10007)\nexcept RuntimeError as e:\n    assert 'offt_3d_init failed' in str(e), e; print('REFUSED')\n"
            "po = api.offt_3d_init(8, 8, 5000); api.offt_3d_fin(po); print('OK5000')\n" % (ROOT, os.path.join(ROOT, "tests")))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OFFT_BLUESTEIN_LONG="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = p.stdout.decode()
    assert p.returncode == 0 and "REFUSED" in out and "OK5000" in out, out   # (5000 = 2^3 5^4 fits the any-length kernel)


def test_linearity_and_shift(built):
    n = 64
    f, g = O.hash_field(n, n, n), O.hash_field(n, n, n, 7, 11, 13)
    F, _ = gpu_fft((n, n, n), field=f)
    Gq, _ = gpu_fft((n, n, n), field=g)
    H, _ = gpu_fft((n, n, n), field=2.5 * f - 1j * g)
    assert rel(H, 2.5 * F - 1j * Gq) < TOL64
    # circular shift theorem along z
    Fs, _ = gpu_fft((n, n, n), field=np.roll(f, 3, axis=2))
    k = np.arange(n)
    assert rel(Fs, F * np.exp(-2j * np.pi * 3 * k / n)[None, None, :]) < TOL64


@pytest.mark.parametrize("streams", ["1", "2"])
def test_y_x_launches_alternating_over_small_plane_groups(built, monkeypatch, streams):
    """the forward z-y-x schedule of one GPU with its y and x launches alternating over groups of z-planes
    (execute_single; by default 256 MiB groups, here 1 MiB: one to four planes per group, ragged last groups, hundreds of
    launch pairs), x launches on the same or on a second stream; double and single precision (column-pair kernels with
    cache-keeping stores), complex and real input"""
    monkeypatch.setenv("OFFT_ZGROUP_MIB", "1")
    monkeypatch.setenv("OFFT_ZGROUP_STREAMS", streams)
    for shape in ((256, 256, 131), (128, 128, 77), (512, 512, 5)):
        got, _ = gpu_fft(shape)
        check64(got, np.fft.fftn(O.hash_field(*shape)))
    shape = (256, 256, 67)
    got, _ = gpu_fft(shape, precision=api.F32)
    want = np.fft.fftn(O.hash_field(*shape).astype(np.complex64).astype(np.complex128))
    assert rel(got.astype(np.complex128), want) <= TOL32
    shape = (256, 256, 90)
    got, _ = gpu_fft(shape, is_r2c=1)
    check64(got, np.fft.rfftn(O.hash_field(*shape).real))


def test_forced_tile_pipeline_on_one_gpu(built, monkeypatch):
    """the multi-rank code path (tile ring, fused pack/unpack descriptors, streams/events) with p = 1"""
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    for shape, kw in [((64, 64, 64), dict(T1=4, W1=2)), ((32, 16, 64), dict(T1=5, W1=1, S=1)), ((20, 12, 18), dict(T1=3, W1=0)),
                      ((128, 128, 128), dict()), ((64, 64, 64), dict(T1=24, T2=5)), ((256, 32, 64), dict(T1=64, T2=16)),
                      ((20, 12, 18), dict(T1=3, T2=4))]:
        got, _ = gpu_fft(shape, **kw)
        want, _, _ = O.world_fft(*shape, 1, kind=1, **kw)
        check64(got, want)


def test_async_steps_and_wait(built, monkeypatch):
    """offt_hip_set_async + offt_hip_wait: several transforms enqueued back to back (what bench.py times on several
    ranks) give bit-identical results to the same number of synchronous calls -- direct path and tile pipelines"""
    L = api.lib()
    for env, shape, kw in ((None, (64, 64, 64), {}), ("1", (64, 32, 128), dict(T1=8, T2=8)), ("1", (64, 64, 32), dict(T1=16, T2=4, S=1))):
        if env:
            monkeypatch.setenv("OFFT_FORCE_PIPELINE", env)
        else:
            monkeypatch.delenv("OFFT_FORCE_PIPELINE", raising=False)
        res = []
        for asyn in (0, 1):
            po = api.offt_3d_init(*shape, custom_params=api.make_params(**kw))
            L.offt_hip_set_output_scale(po, 2.0 ** -9)
            dev, _ = make_input(po, O.hash_field(*shape))
            L.offt_hip_set_async(po, asyn)
            for _ in range(3):
                api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            assert L.offt_hip_wait(po) == 0
            res.append(dev.cpu().numpy().copy())
            api.offt_3d_fin(po)
        assert np.array_equal(res[0], res[1]), (shape, kw)


def test_host_pointer_boundary(built):
    """the reference hands over a calloc'ed host array (run-fft.c:304): staged through HBM"""
    n = 32
    po = api.offt_3d_init(n, n, n)
    f = O.hash_field(n, n, n)
    buf = f.ravel().copy()  # in-place transform: keep f intact for the check
    p = buf.ctypes.data_as(C.c_void_p)
    api.offt_3d_execute(po, p, p)
    c = api.comm_dict(po)
    api.offt_3d_fin(po)
    s0, s1, s2 = c["ostride"]
    idx = (np.arange(n)[:, None, None] * s0 + np.arange(n)[None, :, None] * s1 + np.arange(n)[None, None, :] * s2).ravel()
    check64(buf[idx].reshape(n, n, n), np.fft.fftn(f))


def test_sweep_variants_agree(built):
    L = api.lib()
    n = 1024
    shape = (n, 8, 8)
    base, _ = gpu_fft(shape)
    nv = L.offt_hipk_variant_count(n, 0)
    assert nv >= 2
    for v in range(nv):
        po = api.offt_3d_init(*shape)
        for ax in range(3):
            L.offt_hip_set_variant(po, ax, v)
        dev, _ = make_input(po, O.hash_field(*shape))
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        got = read_output(po, dev, shape)
        api.offt_3d_fin(po)
        assert rel(got, base) < 1e-15 * 10


def test_rccl_self_exchange_one_rank(built, monkeypatch):
    """RCCL binding, communicator split, grouped send/recv and the stream/event ring, exercised with a
    one-rank communicator (all the 1-GPU box allows): every tile goes through ncclSend/ncclRecv to self."""
    L = api.lib()
    uid = (C.c_char * 128)()
    assert L.offt_hip_get_unique_id(uid) == 0, L.offt_hip_last_error()
    assert L.offt_hip_set_world(0, 1, uid, 0) == 0, L.offt_hip_last_error()
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    monkeypatch.setenv("OFFT_FORCE_A2A", "1")
    try:
        for shape, kw in [((64, 64, 64), dict(T1=8, W1=2)), ((128, 64, 32), dict(T1=16, W1=1, S=1)), ((20, 12, 18), dict(T1=3, W1=0))]:
            got, _ = gpu_fft(shape, **kw)
            want, _, _ = O.world_fft(*shape, 1, kind=1, **kw)
            check64(got, want)
    finally:
        L.offt_hip_finalize_world()


@pytest.mark.parametrize("shape", [(16, 16, 16), (64, 64, 64), (128, 32, 256), (32, 32, 1024), (20, 12, 18), (8, 8, 2)])
def test_r2c(built, shape):
    """real-to-complex (-R): z pass real -> Nz/2+1 complex, then y and x on the half spectrum"""
    for layout in (dict(S=1), dict()):
        got, c = gpu_fft(shape, is_r2c=1, **layout)
        want, comms, _ = O.world_fft(*shape, 1, kind=1, is_r2c=1, **layout)
        check64(got, want)
        assert c["ostride"] == comms[0]["ostride"] and c["istride"] == comms[0]["istride"]
    got, _ = gpu_fft(shape, is_r2c=1, precision=api.F32)
    assert rel(got.astype(np.complex128), np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2))) < TOL32


def test_r2c_forced_pipeline(built, monkeypatch):
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    for shape, kw in [((64, 64, 64), dict(T1=8, W1=2)), ((32, 16, 30), dict(T1=5, W1=1, S=1))]:
        got, _ = gpu_fft(shape, is_r2c=1, **kw)
        want, _, _ = O.world_fft(*shape, 1, kind=1, is_r2c=1, **kw)
        check64(got, want)


def test_mpi_harness_runs_one_rank(built, tmp_path):
    """the MPI build of the harness (rank/size from MPI, RCCL id by MPI_Bcast: the reference's launch model,
    run-fft.c:158-160) under mpiexec with the one rank a one-GPU box allows: same stdout lines and spot values"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists("/opt/conda/include/mpi.h") or not os.path.exists(mpiexec):
        pytest.skip("no MPI in this image")
    exe = tmp_path / "run-fft-mpi"
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-DOFFT_HARNESS_MPI", "-I" + os.path.join(root, "include"), "-I/opt/conda/include",
                           "-o", str(exe), os.path.join(root, "harness", "run-fft.c"), "-L" + os.path.join(root, "offt_amd"), "-loffthip",
                           "-L/opt/rocm/lib", "-lamdhip64", "/opt/conda/lib/libmpi.so", "-lm", "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu",
                           "-Wl,-rpath," + os.path.join(root, "offt_amd"), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/conda/lib"])
    n = 128
    out = subprocess.check_output([mpiexec, "-n", "1", str(exe), "-N", str(n), "-n", str(n), "-L", str(n), "-r", "2", "-v"],
                                  stderr=subprocess.STDOUT, timeout=300).decode()
    lines = out.splitlines()
    assert any(l.startswith("@ FINAL") for l in lines) and any(l.startswith("t_min ") for l in lines), out
    spots = [l for l in lines if l.startswith("p 0: 0 0 ")]
    assert len(spots) == 4, out
    assert float(spots[0].split(":")[2].split()[0]) == n ** 3 * 111 * (n - 1) / 2


def test_harness_and_static_sweep(built, tmp_path):
    """bin/run-fft (run-fft.c counterpart): reference stdout lines, closed-form spot values, and -l N = the
    static sweep that replaces the Active-Harmony tuner, logging the reference's "perf v0..v23" point lines"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "bin", "run-fft")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", root, "harness"])
    db = tmp_path / "points.db"
    env = dict(os.environ, OFFT_SWEEP_DB=str(db))
    n = 512
    out = subprocess.check_output([exe, "-N", str(n), "-n", str(n), "-L", str(n), "-r", "2", "-v", "-g", "-l", "4"],
                                  env=env, stderr=subprocess.STDOUT).decode()
    lines = out.splitlines()
    assert any(l.startswith("@ INPUT") for l in lines) and any(l.startswith("@ FINAL") for l in lines)
    assert any(l.startswith("t_init ") for l in lines) and any(l.startswith("t_min ") for l in lines)
    sweep = [l for l in lines if l.startswith("@ SWEEP")]
    assert len(sweep) >= 2, out
    spots = [l for l in lines if l.startswith("p 0: 0 0 ")]
    assert len(spots) == 4
    x0 = float(spots[0].split(":")[2].split()[0])
    assert x0 == n ** 3 * 111 * (n - 1) / 2
    for k in (1, 2, 3):
        re, im = map(float, spots[k].split(":")[2].split())
        cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n))
        assert abs(complex(re, im) - cf) / abs(cf) < 1e-9  # printed with 5 decimals
    pts = [l.split() for l in open(db).read().splitlines()]
    assert len(pts) == len(sweep) and all(len(p) == 25 for p in pts)  # perf + 24 parameters
    final = [l for l in lines if l.startswith("@ FINAL")][0].split()
    # the first point is the registry default; a uniform variant replaces it only when it is at least 2 % faster
    chosen = [p for p in pts if int(p[1 + 3]) == int(final[final.index("Px1") + 1]) and int(p[1 + 4]) == int(final[final.index("Py1") + 1])]
    assert len(chosen) == 1
    if chosen[0] is not pts[0]:
        assert float(chosen[0][0]) <= 0.985 * float(pts[0][0])
    else:
        assert all(float(p[0]) >= 0.975 * float(pts[0][0]) for p in pts[1:])
    # second run: every point is already in the database and is not re-timed
    out2 = subprocess.check_output([exe, "-N", str(n), "-n", str(n), "-L", str(n), "-l", "4"], env=env,
                                   stderr=subprocess.STDOUT).decode()
    assert len(open(db).read().splitlines()) == len(pts), out2
    # through the MPI-free launcher (one rank here: the box has one GPU): tools/launch.py -n 1 -- bin/run-fft ...
    out3 = subprocess.check_output([sys.executable, os.path.join(root, "tools", "launch.py"), "-n", "1", "--", exe, "-N", "64", "-n", "64",
                                    "-L", "64", "-v"], stderr=subprocess.STDOUT, timeout=300).decode()
    assert any(l.startswith("p 0: 0 0 0: ") for l in out3.splitlines()), out3
    # a run the library cannot do must FAIL: non-zero exit and the reference's t_min 999999999 line, not timings of
    # an untransformed buffer (10007 points, a prime, with the Bluestein net under the four-step path switched off: no kernel)
    p = subprocess.run([exe, "-N", "10007", "-n", "4", "-L", "4"], env=dict(env, OFFT_BLUESTEIN_LONG="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode != 0 and "t_min 999999999" in p.stdout.decode(), p.stdout.decode()
