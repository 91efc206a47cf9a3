"""CPU tests of the PRODUCT's host logic (offt_host.c) with the GPU operations swapped for the
test-only CPU interpreter: decomposition, defaults, pass descriptors, tile ring, exchange schedule.
Multi-rank cases run as real processes over `gloo` (world_size 2 and 4)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cpu_world
import oracle_lib as O
from offt_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-14


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture()
def cpu1(built):
    cpu_world.install(0, 1, p1=1)
    yield
    cpu_world.uninstall()


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 8, 4), (6, 10, 9), (1, 5, 1), (32, 2, 3)])
@pytest.mark.parametrize("layout", [dict(S=1), dict(), dict(eq=1)])
def test_single_rank_direct(cpu1, shape, layout):
    eq = layout.get("eq", 0)
    if eq and shape[0] != shape[1]:
        pytest.skip("y-z-x layout needs Nx == Ny (offt.h:160)")
    params = {k: v for k, v in layout.items() if k != "eq"}
    c, v, buf = cpu_world.run_rank(*shape, is_equalxy=eq, **params)
    G = np.zeros(shape, dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL
    # same decomposition and defaults as the restated reference
    oc = O.comm(*shape, 1, 0, 1, 0, eq, params.get("S", 0))
    for k in oc:
        if k in c:
            assert c[k] == oc[k], k
    dv = O.params_default(*shape, 1)
    for i, n in enumerate(O.PARAM_NAMES):
        if n not in params:
            assert v[i] == dv[i], n


def test_single_rank_forced_pipeline(cpu1, monkeypatch):
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    for shape, kw in [((8, 8, 8), dict(T1=2, W1=1)), ((12, 6, 10), dict(T1=5, W1=2)), ((8, 8, 8), dict(T1=3, W1=0, S=1)),
                      ((4, 4, 4), dict(T1=100)), ((12, 6, 10), dict(T1=5, T2=3)), ((9, 4, 7), dict(T1=2, T2=1))]:
        c, v, buf = cpu_world.run_rank(*shape, **kw)
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL, (shape, kw)


def test_forced_self_exchange(cpu1, monkeypatch):
    """p = 1 with the exchanges forced on: separate send/receive buffers and the a2a callback in the loop"""
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    monkeypatch.setenv("OFFT_FORCE_A2A", "1")
    for shape, kw in [((8, 8, 8), dict(T1=2, W1=1)), ((6, 10, 4), dict(T1=4, W1=2, S=1))]:
        c, v, buf = cpu_world.run_rank(*shape, **kw)
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL, (shape, kw)


@pytest.mark.parametrize("shape", [(8, 8, 8), (4, 6, 10), (16, 4, 7)])
@pytest.mark.parametrize("layout", [dict(S=1), dict()])
def test_r2c_single_rank(cpu1, shape, layout):
    """real-to-complex z pass (-R, offt-compute.c:334-336, 960-961; run-fft.c:53-54)"""
    c, v, buf = cpu_world.run_rank(*shape, is_r2c=1, **layout)
    G = np.zeros((shape[0], shape[1], shape[2] // 2 + 1), dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2))) < TOL
    oc = O.comm(*shape, 1, 0, 1, 1, 0, layout.get("S", 0))
    for k in oc:
        if k in c:
            assert c[k] == oc[k], k


def test_inverse_roundtrip_single(cpu1):
    for shape, kw in [((8, 4, 16), dict(S=1)), ((8, 8, 6), dict())]:
        cp = api.make_params(**kw)
        po = api.offt_3d_init(*shape, custom_params=cp)
        c = api.comm_dict(po)
        f = O.hash_field(*shape)
        buf = np.zeros(api.local_elems(po), dtype=np.complex128)
        s0, s1, s2 = c["istride"]
        idx = (np.arange(shape[0])[:, None, None] * s0 + np.arange(shape[1])[None, :, None] * s1 + np.arange(shape[2])[None, None, :] * s2).ravel()
        buf[idx] = f.ravel()
        import ctypes as C
        p = buf.ctypes.data_as(C.c_void_p)
        api.offt_3d_execute_dir(po, p, p, -1)
        api.offt_3d_execute_dir(po, p, p, +1)
        api.offt_3d_fin(po)
        assert rel(buf[idx].reshape(shape) / np.prod(shape), f) < TOL


def test_defaults_match_oracle_many(cpu1):
    for N, p in [((128, 128, 128), 2), ((512, 512, 512), 1), ((1024, 1024, 1024), 1), ((1024, 1024, 1024), 8),
                 ((2048, 2048, 2048), 8), ((100, 60, 36), 6), ((17, 33, 5), 4)]:
        cpu_world.install(0, p, p1=None)
        L = api.lib()
        # only the parameter / decomposition part of init: a world of p ranks as rank 0 without exchanging
        cp = api.make_params()
        # big grids would allocate host tile buffers in the CPU backend: use the oracle-checked default P1 and skip alloc
        dv = O.params_default(*N, p)
        if np.prod(N) > 1 << 22:
            continue
        po = api.offt_3d_init(*N, custom_params=cp)
        assert list(po.contents.params.contents.v) == dv, (N, p)
        oc = O.comm(*N, p, 0, dv[0])
        c = api.comm_dict(po)
        for k in oc:
            if k in c:
                assert c[k] == oc[k], (N, p, k)
        api.offt_3d_fin(po)
    cpu_world.install(0, 1, p1=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_world(size, cases, tmp_path):
    port = _free_port()
    procs = []
    for r in range(size):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"),
                                       json.dumps(cases), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r]}"
    for ci, case in enumerate(cases):
        shape = tuple(case["N"])
        r2c = case.get("r2c", 0)
        oshape = (shape[0], shape[1], shape[2] // 2 + 1) if r2c else shape
        G = np.full(oshape, np.nan + 0j)
        for r in range(size):
            meta = json.load(open(tmp_path / f"case{ci}_rank{r}.json"))
            buf = np.load(tmp_path / f"case{ci}_rank{r}.npy")
            cpu_world.scatter_out(meta["comm"], buf, G)
            oc = O.comm(*shape, size, r, meta["v"][0], r2c, case.get("eq", 0), meta["v"][23])
            for k in oc:
                if k in meta["comm"]:
                    assert meta["comm"][k] == oc[k], (case, r, k)
        assert not np.isnan(G).any(), case
        want = np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2)) if r2c else np.fft.fftn(O.hash_field(*shape))
        assert rel(G, want) < TOL, case
        # and against the restated reference pipeline on the same decomposition
        og, _, _ = O.world_fft(*shape, size, kind=1, is_equalxy=case.get("eq", 0), is_r2c=r2c, **case["params"])
        assert rel(G, og) < TOL, case
        if case.get("inv"):  # offt_3d_execute_dir(+1) on the result gives back every rank's input block * N
            for r in range(size):
                meta = json.load(open(tmp_path / f"case{ci}_rank{r}.json"))["comm"]
                back = np.load(tmp_path / f"case{ci}_rank{r}_inv.npy")
                if back.size == 0:
                    continue
                blk = O.hash_field(*meta["isize"], *meta["istart"])
                assert rel(back / np.prod(shape), blk) < TOL, (case, r)


def test_gloo_world2(built, tmp_path):
    cases = [dict(N=[8, 8, 8], params=dict(P1=1)), dict(N=[8, 8, 8], params=dict(P1=2)),
             dict(N=[8, 8, 8], params=dict(P1=1, S=1, T1=2, W1=1)), dict(N=[8, 8, 8], params=dict(P1=2), eq=1),
             dict(N=[9, 7, 5], params=dict(P1=2, T1=2, W1=2)), dict(N=[9, 7, 5], params=dict(P1=1, T1=4, W1=0, S=1)),
             dict(N=[16, 16, 4], params=dict()),
             dict(N=[8, 8, 8], params=dict(P1=1), r2c=1), dict(N=[8, 6, 12], params=dict(P1=2, S=1), r2c=1),
             # slab schedule with several z-chunks and a ragged last x-tile / z-chunk
             dict(N=[10, 6, 9], params=dict(P1=1, T1=3, T2=2)), dict(N=[16, 8, 16], params=dict(P1=1, T1=8, T2=1)),
             dict(N=[7, 5, 11], params=dict(P1=1, T1=2, T2=4), r2c=1),
             # multi-rank inverse (extension): slab and pencil schedules replayed backwards
             dict(N=[8, 8, 8], params=dict(P1=1), inv=1), dict(N=[10, 6, 9], params=dict(P1=1, T1=3, T2=2), inv=1),
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, W1=1), inv=1), dict(N=[9, 7, 5], params=dict(P1=2, S=1), inv=1)]
    _run_world(2, cases, tmp_path)


def test_gloo_world4(built, tmp_path):
    cases = [dict(N=[8, 8, 8], params=dict(P1=2)), dict(N=[8, 8, 8], params=dict(P1=4, S=1)),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=3)), dict(N=[10, 6, 9], params=dict(P1=2, T1=2, W1=1)),
             dict(N=[6, 10, 7], params=dict(P1=2, S=1))]
    _run_world(4, cases, tmp_path)


def test_gloo_world8(built, tmp_path):
    """the 8-rank shapes the multi-GPU bench uses (1x8 slab, 2x4 pencil, 8x1), incl. r2c and ragged sizes"""
    cases = [dict(N=[16, 16, 16], params=dict(P1=1)), dict(N=[16, 16, 16], params=dict()),
             dict(N=[16, 16, 16], params=dict(P1=8)), dict(N=[16, 16, 16], params=dict(P1=1, S=1, T1=4)),
             dict(N=[20, 12, 18], params=dict(P1=2, T1=3, W1=1)), dict(N=[16, 16, 16], params=dict(P1=1), r2c=1),
             dict(N=[16, 16, 16], params=dict(P1=4), eq=1), dict(N=[16, 16, 16], params=dict(P1=1), inv=1),
             dict(N=[16, 16, 16], params=dict(), inv=1)]
    _run_world(8, cases, tmp_path)


def test_baseline_config0_128cube_two_ranks(built, tmp_path):
    """BASELINE.json configs[0]: 128^3 double-complex on 2 ranks, CPU plumbing (no GPU): the product's host
    pipeline over gloo with the reference's default parameters (P1=1, T1=8, W1=2 -- survey_recorded.json)."""
    _run_world(2, [dict(N=[128, 128, 128], params=dict())], tmp_path)
    import json
    meta = json.load(open(tmp_path / "case0_rank0.json"))
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_recorded.json")))
    assert {n: meta["v"][i] for i, n in enumerate(O.PARAM_NAMES)} == rec["default_params_N128_p2"]
