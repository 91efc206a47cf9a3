"""Check the oracle (CPU restatement of offt-compute.c) BEFORE it is used as the checker.

PARITY UNPINNED vs a reference build: the reference has no tests or golden vectors and cannot be built here (no FFTW).
Numerics are pinned by numpy.fft (pocketfft) full grids and the closed-form ramp; the decomposition, default
parameters and layouts by restating offt-compute.c, corroborated by values the survey stage recorded from a run of the
reference linked against stand-in FFTW headers + MKL (tests/golden/ref_*.npz, survey_recorded.json -- a stand-in
build, kept as corroboration, not as reference truth)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-14  # double, N <= 32: a few ulp * log2(N)


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_fft1d_any_length():
    rng = np.random.default_rng(1)
    for n in [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 18, 20, 25, 27, 30, 32, 49, 64, 97, 100, 128, 243, 256, 1000, 1024, 2048]:
        a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert rel(O.fft1d(a), np.fft.fft(a)) < 2e-15 * max(1.0, np.log2(n)), n


def test_reference_dump_18cube_6ranks():
    """Full-grid output of the survey's STAND-IN build of the reference (declarations-only FFTW headers + MKL):
    N=18, 6 ranks, p1=2, S=1 -- corroboration, not a pin."""
    d = np.load(os.path.join(G, "ref_n18_p6_p1-2_S1.npz"))
    g, comms, v = O.world_fft(18, 18, 18, 6, kind=1, P1=2, S=1)
    ref = np.full((18, 18, 18), np.nan + 0j)
    rxyz = d["rank_xyz"].astype(int)
    ref[rxyz[:, 1], rxyz[:, 2], rxyz[:, 3]] = d["value"]
    assert not np.isnan(ref).any()
    assert rel(g, ref) < 1e-15
    assert np.abs(g - ref).max() < 1e-12
    # layout contract (ostart, osize, ostride per rank) as printed by the reference
    for r in range(6):
        mine = comms[r]["ostart"] + comms[r]["osize"] + comms[r]["ostride"]
        assert mine == list(d["ostart_osize_ostride"][r]), r
    # and every point was owned by exactly the rank the reference said
    for r in range(6):
        c = comms[r]
        sel = rxyz[rxyz[:, 0] == r][:, 1:]
        assert len(sel) == c["osize"][0] * c["osize"][1] * c["osize"][2]
        assert sel[:, 1].min() == c["ostart"][1] and sel[:, 2].min() == c["ostart"][2]


def test_survey_recorded_values():
    rec = json.load(open(os.path.join(G, "survey_recorded.json")))
    v = O.params_default(128, 128, 128, 2)
    want = rec["default_params_N128_p2"]
    assert {n: v[i] for i, n in enumerate(O.PARAM_NAMES)} == want
    assert O.params_default(1024, 1024, 1024, 8)[0] == rec["default_P1"]["8"]
    assert O.params_default(128, 128, 128, 2)[0] == rec["default_P1"]["2"]
    lay = rec["layout_1024_p8_2x4"]
    c = O.comm(1024, 1024, 1024, 8, 0, 2)
    assert [c["M1"], c["M2"], c["M3"], c["M4"]] == [lay["M1"], lay["M2"], lay["M3"], lay["M4"]]
    assert c["istride"] == lay["istride"] and c["ostride"] == lay["ostride_zyx"]


@pytest.mark.parametrize("n", [8, 16, 20])
def test_numpy_goldens(n):
    F = np.load(os.path.join(G, f"numpy_fftn_hash_{n}.npz"))["F"]
    for p, p1, kw in [(1, 1, {}), (2, 1, {}), (4, 2, dict(S=1)), (4, 2, {}), (4, 4, dict(V=3))]:
        g, _, _ = O.world_fft(n, n, n, p, kind=1, P1=p1, **kw)
        assert rel(g, F) < TOL, (n, p, p1, kw)


# (ranks, N, p1, equalxy, S, V, is_oned): the survey's verified set + the layouts MKL could not run
CASES = [(1, 16, 1, 1, 0, 0, 0), (2, 16, 1, 1, 0, 0, 0), (2, 16, 2, 1, 0, 0, 0), (4, 16, 2, 1, 0, 0, 0),
         (8, 16, 2, 1, 0, 0, 0), (8, 16, 4, 0, 1, 0, 0), (8, 20, 2, 0, 1, 0, 0), (6, 18, 2, 0, 1, 0, 0),
         (4, 16, 2, 0, 0, 0, 0), (6, 18, 3, 0, 0, 0, 0), (6, 20, 2, 0, 0, 0, 0), (8, 20, 2, 0, 1, 3, 0),
         (6, 18, 2, 0, 0, 3, 0), (4, 12, 1, 0, 0, 0, 1), (4, 12, 4, 0, 0, 0, 1), (4, 12, 1, 1, 0, 0, 1),
         (4, 12, 4, 1, 0, 0, 1), (4, 12, 1, 0, 1, 0, 1), (4, 12, 4, 0, 1, 0, 1), (3, 10, 1, 0, 0, 3, 1),
         (5, 7, 5, 0, 0, 0, 0), (7, 5, 1, 0, 1, 2, 0)]


@pytest.mark.parametrize("case", CASES)
def test_world_matches_fftn(case):
    p, n, p1, eq, S, V, oned = case
    g, comms, v = O.world_fft(n, n, n, p, kind=1, is_equalxy=eq, is_oned=oned, P1=p1, S=S, V=V)
    assert rel(g, np.fft.fftn(O.hash_field(n, n, n))) < TOL


def test_non_cubic_and_ragged():
    for (nx, ny, nz, p, kw) in [(12, 20, 18, 6, dict(P1=2)), (12, 20, 18, 6, dict(P1=3, S=1, V=3)),
                                (5, 9, 7, 4, dict(P1=2)), (3, 3, 3, 8, dict(P1=2, S=1)), (1, 4, 6, 2, dict(P1=1)),
                                (20, 20, 12, 6, dict(P1=2))]:
        eq = 1 if (nx == ny and kw.get("P1") == 2 and nx == 20) else 0
        g, _, _ = O.world_fft(nx, ny, nz, p, kind=1, is_equalxy=eq, **kw)
        assert rel(g, np.fft.fftn(O.hash_field(nx, ny, nz))) < TOL, (nx, ny, nz, p, kw)


def test_ramp_closed_form():
    """run-fft.c's ramp input: X[0,0,0] = N^3*111*(N-1)/2, X[0,0,k] = N^3(-1/2 + (i/2)cot(pi k/N))."""
    for n in (8, 16, 20, 32):
        g, _, _ = O.world_fft(n, n, n, 2, kind=0, P1=1, is_equalxy=1)
        assert g[0, 0, 0] == n ** 3 * 111 * (n - 1) / 2
        k = np.arange(1, 4)
        cf = n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k / n))
        assert np.abs(g[0, 0, 1:4] - cf).max() / np.abs(cf).max() < 1e-14


def test_r2c_z_pass():
    for n, p, p1 in [(8, 1, 1), (16, 2, 1), (12, 4, 2)]:
        g, _, _ = O.world_fft(n, n, n, p, kind=1, is_r2c=1, P1=p1, S=1)
        f = O.hash_field(n, n, n).real
        assert rel(g, np.fft.rfftn(f, axes=(0, 1, 2))) < TOL


def test_tiles_and_windows_do_not_change_results():
    base, _, _ = O.world_fft(16, 16, 16, 4, kind=1, P1=2)
    for kw in (dict(T1=1, T2=1), dict(T1=3, T2=5, W1=0, W2=0), dict(T1=8, T2=8, Ry=0), dict(Ry=10), dict(T1=16, T2=16)):
        g, _, _ = O.world_fft(16, 16, 16, 4, kind=1, P1=2, **kw)
        assert np.array_equal(g, base), kw
