"""CPU tests of the PRODUCT's host logic (offt_host.c) with the GPU operations swapped for the
test-only CPU interpreter: decomposition, defaults, pass descriptors, tile ring, exchange schedule.
Multi-rank cases run as real processes over `gloo` (world_size 2 and 4)."""
import ctypes as C
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


@pytest.mark.parametrize("streams", ["1", "2"])
def test_single_rank_zgroups(cpu1, monkeypatch, streams):
    """forward z-y-x on one rank with the y and x launches alternating over groups of z-planes (execute_single): 1 MiB
    groups = 16 planes of 64 x 64, i.e. groups of 16, 16 and 8 planes here; also the ragged r2c plane count, and the x
    launches on a second stream"""
    monkeypatch.setenv("OFFT_ZGROUP_MIB", "1")
    monkeypatch.setenv("OFFT_ZGROUP_STREAMS", streams)
    shape = (64, 64, 40)
    c, v, buf = cpu_world.run_rank(*shape)
    G = np.zeros(shape, dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL
    def roundtrip_ok(**kw):  # forward, then the inverse on the result: the input block times Nx Ny Nz
        c, v, buf, back = cpu_world.run_rank(*shape, roundtrip=True, **kw)
        s0, s1, s2 = c["istride"]
        idx = (np.arange(shape[0])[:, None, None] * s0 + np.arange(shape[1])[None, :, None] * s1 + np.arange(shape[2])[None, None, :] * s2).ravel()
        assert rel(back[idx].reshape(shape) / np.prod(shape), O.hash_field(*shape)) < TOL
    roundtrip_ok()
    # the other layouts alternate their z and y launches over groups of x-planes (25, 25 and 14 planes here)
    for kw in (dict(S=1), dict(is_equalxy=1)):
        c, v, buf = cpu_world.run_rank(*shape, **kw)
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL
        roundtrip_ok(**kw)
    # a CUBE: every pass has the same batch count, so a wrongly chosen pair of launches would still "match" and alternate
    for kw in (dict(), dict(S=1), dict(is_equalxy=1), dict(is_equalxy=1, rotate=1), dict(S=1, rotate=1)):
        shape = (64, 64, 64)
        kw = dict(kw)
        monkeypatch.setenv("OFFT_ROTATE", str(kw.pop("rotate", 0)))
        c, v, buf = cpu_world.run_rank(*shape, **kw)
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL, kw
        roundtrip_ok(**kw)
    monkeypatch.delenv("OFFT_ROTATE")
    shape = (64, 64, 38)  # r2c: 20 planes of the half spectrum
    c, v, buf = cpu_world.run_rank(*shape, is_r2c=1)
    G = np.zeros((shape[0], shape[1], shape[2] // 2 + 1), dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.rfftn(O.hash_field(*shape).real)) < TOL


@pytest.mark.parametrize("shape", [(8, 8, 8), (16, 8, 4), (6, 10, 9), (32, 2, 3), (12, 12, 10)])
def test_single_rank_xyz_rotating_schedule(cpu1, monkeypatch, shape):
    """x-y-z output (S = 1) through the rotating three-pass schedule (two scratch volumes, the x pass in the middle); the
    library picks it when an x-plane is a whole number of MiB, OFFT_ROTATE=1 forces it for these small grids.
    Forward, inverse round trip, r2c; and the y-z-x layout's rotating schedule (is_equalxy, Nx == Ny) the same way."""
    monkeypatch.setenv("OFFT_ROTATE", "1")
    c, v, buf, back = cpu_world.run_rank(*shape, S=1, roundtrip=True)
    G = np.zeros(shape, dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL
    assert rel(cpu_world.input_block(c, back) / np.prod(shape), O.hash_field(*shape)) < TOL
    c, v, buf = cpu_world.run_rank(*shape, S=1, is_r2c=1)
    G = np.zeros((shape[0], shape[1], shape[2] // 2 + 1), dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    assert rel(G, np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2))) < TOL
    if shape[0] == shape[1]:
        c, v, buf, back = cpu_world.run_rank(*shape, is_equalxy=1, roundtrip=True)
        assert c["ostride"] == [1, shape[0] * shape[2], shape[0]]
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL
        assert rel(cpu_world.input_block(c, back) / np.prod(shape), O.hash_field(*shape)) < TOL
        c, v, buf = cpu_world.run_rank(*shape, is_equalxy=1, is_r2c=1)
        G = np.zeros((shape[0], shape[1], shape[2] // 2 + 1), dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2))) < TOL


def test_single_rank_forced_pipeline(cpu1, monkeypatch):
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    for shape, kw in [((8, 8, 8), dict(T1=2, W1=1)), ((12, 6, 10), dict(T1=5, W1=2)), ((8, 8, 8), dict(T1=3, W1=0, S=1)),
                      ((4, 4, 4), dict(T1=100)), ((12, 6, 10), dict(T1=5, T2=3)), ((9, 4, 7), dict(T1=2, T2=1))]:
        c, v, buf = cpu_world.run_rank(*shape, **kw)
        G = np.zeros(shape, dtype=np.complex128)
        cpu_world.scatter_out(c, buf, G)
        assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL, (shape, kw)


def test_plan_options_and_exchange_setter(cpu1, monkeypatch):
    """offt_hip_set_option / offt_hip_get_option / offt_hip_set_exchange (include/offt_hip.h): options are plan state, read
    from the environment only at offt_3d_init; the ones that shape buffers rebuild them; the transform stays right"""
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    monkeypatch.setenv("OFFT_ZGROUP_MIB", "7")
    L = api.lib()
    shape = (12, 6, 10)
    po = api.offt_3d_init(*shape, custom_params=api.make_params(T1=5, T2=3))
    OPT = dict(ZGROUP_MIB=0, ZGROUP_STREAMS=1, SLAB_CHUNK_MIB=2, COMM_STREAMS=3, F32_PAIRS=4, K1_STREAMS=5, SELF_BYPASS=6, MIN_MSG=7,
               EXEC_TIMEOUT_S=8, P2P_TIMEOUT_S=9)
    assert L.offt_hip_get_option(po, OPT["ZGROUP_MIB"]) == 7          # the environment's value became the plan's default ...
    monkeypatch.setenv("OFFT_ZGROUP_MIB", "99")
    assert L.offt_hip_get_option(po, OPT["ZGROUP_MIB"]) == 7          # ... and later changes of the environment do not reach the plan
    for name, val in (("ZGROUP_MIB", 3), ("ZGROUP_STREAMS", 2), ("SLAB_CHUNK_MIB", 1), ("COMM_STREAMS", 2), ("F32_PAIRS", 0), ("K1_STREAMS", 2),
                      ("SELF_BYPASS", 0), ("MIN_MSG", 0), ("EXEC_TIMEOUT_S", 17), ("P2P_TIMEOUT_S", 5)):
        assert L.offt_hip_set_option(po, OPT[name], val) == 0, name
        assert L.offt_hip_get_option(po, OPT[name]) == val, name
    assert L.offt_hip_set_option(po, 12345, 1) != 0                   # unknown option
    assert L.offt_hip_get_exchange(po) == 0
    assert L.offt_hip_set_exchange(po, 1) == 0                        # a world of one: nothing to map, the staged exchange stays
    c = api.comm_dict(po)
    buf = np.zeros(api.local_elems(po), dtype=np.complex128)
    s0, s1, s2 = c["istride"]
    idx = (np.arange(shape[0])[:, None, None] * s0 + np.arange(shape[1])[None, :, None] * s1 + np.arange(shape[2])[None, None, :] * s2).ravel()
    buf[idx] = O.hash_field(*shape).ravel()
    ptr = buf.ctypes.data_as(C.c_void_p)
    api.offt_3d_execute(po, ptr, ptr)
    G = np.zeros(shape, dtype=np.complex128)
    cpu_world.scatter_out(c, buf, G)
    api.offt_3d_fin(po)
    assert rel(G, np.fft.fftn(O.hash_field(*shape))) < TOL


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
        oparams = dict(case["params"])
        if case.get("max_loop"):  # the static sweep chose the mesh: the oracle runs on the mesh the plan ended up with
            oparams["P1"] = json.load(open(tmp_path / f"case{ci}_rank0.json"))["v"][0]
        og, _, _ = O.world_fft(*shape, size, kind=1, is_equalxy=case.get("eq", 0), is_r2c=r2c, **oparams)
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
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, W1=1), inv=1), dict(N=[9, 7, 5], params=dict(P1=2, S=1), inv=1),
             # pencil schedule, phase 2 in z-chunks of T2 planes (offt-compute.c:3682-3862): several chunks, a ragged
             # last chunk, W2 = 0 (no overlap), every output layout, r2c, inverse replay
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, W1=1, T2=2)), dict(N=[9, 7, 11], params=dict(P1=2, T1=2, T2=3)),
             dict(N=[9, 7, 11], params=dict(P1=2, T1=4, T2=4, W2=0, S=1)), dict(N=[8, 8, 10], params=dict(P1=2, T2=3), eq=1),
             dict(N=[8, 6, 12], params=dict(P1=2, T1=3, T2=2), r2c=1), dict(N=[9, 7, 11], params=dict(P1=2, T1=2, T2=3), inv=1),
             dict(N=[9, 7, 11], params=dict(P1=1, T1=2, T2=3, S=1)),
             # both exchange-volume layouts of each schedule on the same even grid: contiguous lines (default) and the
             # first version (forced), incl. r2c, inverse and the x-y-z / y-z-x output layouts
             dict(N=[8, 8, 14], params=dict(P1=2, T1=2, T2=4), r2c=1), dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=2, S=1)),
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=4), eq=1), dict(N=[8, 8, 8], params=dict(P1=2, T1=4, T2=2), inv=1),
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=2), env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=2), inv=1, env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=2, T2=2), env=dict(OFFT_SLAB_XC_LAYOUT=1)),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=2, T2=2), inv=1),
             # the defaults' own tiling (no 4 MiB message floor): T1 = M1/16, T2 = M3/16 merged to <= 8 chunks
             dict(N=[32, 16, 32], params=dict(P1=2), env=dict(OFFT_MIN_MSG=0)),
             dict(N=[32, 16, 32], params=dict(P1=1), env=dict(OFFT_MIN_MSG=0)),
             # the self block sent through the exchange like any other (OFFT_SELF_BYPASS=0; the default stores it straight
             # into the receive side and tests/cpu_world.py asserts that no exchange carries it)
             dict(N=[8, 8, 8], params=dict(P1=1, T1=2, T2=2), env=dict(OFFT_SELF_BYPASS=0)),
             dict(N=[9, 7, 11], params=dict(P1=2, T1=2, T2=3), inv=1, env=dict(OFFT_SELF_BYPASS=0))]
    _run_world(2, cases, tmp_path)


def test_gloo_world4(built, tmp_path):
    cases = [dict(N=[8, 8, 8], params=dict(P1=2)), dict(N=[8, 8, 8], params=dict(P1=4, S=1)),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=3)), dict(N=[10, 6, 9], params=dict(P1=2, T1=2, W1=1)),
             dict(N=[6, 10, 7], params=dict(P1=2, S=1)),
             dict(N=[10, 6, 9], params=dict(P1=2, T1=2, W1=1, T2=2)), dict(N=[12, 8, 10], params=dict(P1=4, T1=1, T2=3)),
             dict(N=[10, 6, 9], params=dict(P1=2, T1=2, T2=2), inv=1), dict(N=[16, 16, 16], params=dict(), env=dict(OFFT_MIN_MSG=0)),
             dict(N=[16, 8, 12], params=dict(P1=2, T1=2, T2=3)), dict(N=[16, 8, 12], params=dict(P1=4, T1=2, T2=4, S=1), inv=1),
             dict(N=[16, 8, 12], params=dict(P1=2, T1=4, T2=2), env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[16, 8, 12], params=dict(P1=2, T1=4, T2=2), env=dict(OFFT_SELF_BYPASS=0)),
             dict(N=[10, 6, 9], params=dict(P1=1, T1=3, T2=2), inv=1, env=dict(OFFT_SELF_BYPASS=0))]
    _run_world(4, cases, tmp_path)


def test_gloo_world4_static_sweep_with_mesh(built, tmp_path):
    """max_loop > 0 on several ranks: the sweep tries the meshes P1 in {default, 1, p} (rebuilding decomposition,
    buffers and groups per point like offt-tuning.c:929), then tilings; every point is timed as a max over ranks, so
    all ranks must end on the SAME point; rank 0 logs `perf v0..v23` lines (offt-tuning.c:231-277)."""
    cases = [dict(N=[16, 16, 16], params=dict(), max_loop=7, env=dict(OFFT_SWEEP_DB="{outdir}/sweep.db", OFFT_MIN_MSG=0)),
             dict(N=[12, 8, 10], params=dict(P1=2), max_loop=3, env=dict(OFFT_SWEEP_DB="{outdir}/sweep2.db"))]
    _run_world(4, cases, tmp_path)
    vs = [json.load(open(tmp_path / f"case0_rank{r}.json"))["v"] for r in range(4)]
    assert all(v == vs[0] for v in vs), vs
    lines = [ln.split() for ln in open(tmp_path / "sweep.db").read().splitlines()]
    assert len(lines) == 7 and all(len(ln) == 25 for ln in lines)
    assert sorted({int(ln[1]) for ln in lines[:3]}) == [1, 2, 4]            # stage A: the three meshes
    assert len({int(ln[1]) for ln in lines[3:]}) == 1                       # stage B stays on the winner
    assert vs[0][0] == int(lines[3][1])
    vs2 = [json.load(open(tmp_path / f"case1_rank{r}.json"))["v"] for r in range(4)]
    assert all(v == vs2[0] for v in vs2) and vs2[0][0] == 2                 # -d given: the mesh is not swept
    assert {int(ln.split()[1]) for ln in open(tmp_path / "sweep2.db").read().splitlines()} == {2}


def test_gloo_world8_sweep_covers_the_whole_p1_lattice(built, tmp_path):
    """stage A of the static sweep enumerates every P1 the reference's lattice admits (offt-compute.c:3002-3023): on 8
    ranks and 16^3 that is 1, 2, 4 and 8 -- four mesh points, identical choices on all ranks"""
    cases = [dict(N=[16, 16, 16], params=dict(), max_loop=6, env=dict(OFFT_SWEEP_DB="{outdir}/sweep8.db", OFFT_MIN_MSG=0))]
    _run_world(8, cases, tmp_path)
    vs = [json.load(open(tmp_path / f"case0_rank{r}.json"))["v"] for r in range(8)]
    assert all(v == vs[0] for v in vs), vs
    lines = [ln.split() for ln in open(tmp_path / "sweep8.db").read().splitlines()]
    assert len(lines) == 6
    assert [int(ln[1]) for ln in lines[:4]] == [2, 1, 8, 4]                 # default first, the slab shapes, then the rest
    assert len({int(ln[1]) for ln in lines[4:]}) == 1 and vs[0][0] == int(lines[4][1])
    # the winner keeps the tiling it was timed with: its stage-A line reappears as the plan's final T1 / T2 unless stage B beat it
    win = [ln for ln in lines[:4] if int(ln[1]) == vs[0][0]][0]
    best_line = min(lines, key=lambda ln: float(ln[0]))
    assert [int(x) for x in best_line[1:]] == vs[0] or float(best_line[0]) >= 99999999.0, (best_line, vs[0], win)


def test_gloo_world4_one_rank_out_of_memory_during_the_sweep(built, tmp_path):
    """a local failure (here: an allocation) on ONE rank while the sweep rebuilds the mesh must not leave the other ranks
    alone in a collective: the point is marked infeasible by every rank (99999999, offt-compute.c:3881) and the sweep goes on"""
    cases = [dict(N=[16, 16, 16], params=dict(), max_loop=5, fail_alloc=dict(rank=2),
                  env=dict(OFFT_SWEEP_DB="{outdir}/sweep_oom.db", OFFT_MIN_MSG=0))]
    _run_world(4, cases, tmp_path)
    vs = [json.load(open(tmp_path / f"case0_rank{r}.json"))["v"] for r in range(4)]
    assert all(v == vs[0] for v in vs), vs
    lines = [ln.split() for ln in open(tmp_path / "sweep_oom.db").read().splitlines()]
    assert float(lines[1][0]) >= 99999999.0 and float(lines[0][0]) < 99999999.0   # the second mesh point failed, for everybody
    assert vs[0][0] != int(lines[1][1])                                           # ... and was not chosen


def test_gloo_world8(built, tmp_path):
    """the 8-rank shapes the multi-GPU bench uses (1x8 slab, 2x4 pencil, 8x1), incl. r2c and ragged sizes"""
    cases = [dict(N=[16, 16, 16], params=dict(P1=1)), dict(N=[16, 16, 16], params=dict()),
             dict(N=[16, 16, 16], params=dict(P1=8)), dict(N=[16, 16, 16], params=dict(P1=1, S=1, T1=4)),
             dict(N=[20, 12, 18], params=dict(P1=2, T1=3, W1=1)), dict(N=[16, 16, 16], params=dict(P1=1), r2c=1),
             dict(N=[16, 16, 16], params=dict(P1=4), eq=1), dict(N=[16, 16, 16], params=dict(P1=1), inv=1),
             dict(N=[16, 16, 16], params=dict(), inv=1),
             dict(N=[16, 16, 16], params=dict(T1=2, T2=1)), dict(N=[20, 12, 18], params=dict(P1=2, T1=3, W1=1, T2=2)),
             dict(N=[16, 16, 16], params=dict(P1=8, T1=1, T2=4)), dict(N=[16, 16, 16], params=dict(P1=4, T2=2), inv=1)]
    _run_world(8, cases, tmp_path)


def test_exchange_failure_is_reported_not_hidden(built, monkeypatch):
    """an exchange that fails mid-schedule (a dead peer, an RCCL error) must surface as the reference's only failure
    convention, t[ALL] = 99999999 (offt-compute.c:3881), plus offt_hip_last_error() -- never as a normal return"""
    monkeypatch.setenv("OFFT_FORCE_PIPELINE", "1")
    monkeypatch.setenv("OFFT_FORCE_A2A", "1")
    for kw in (dict(T1=2, W1=1, T2=2), dict(T1=2, W1=1, S=1)):
        cpu_world.install(0, 1, fail_after=3)
        try:
            with pytest.raises(RuntimeError, match="offt_3d_execute"):
                cpu_world.run_rank(8, 8, 8, **kw)
        finally:
            cpu_world.uninstall()
    # the C-level view of the same failure: the marker in t[ALL]
    cpu_world.install(0, 1, fail_after=2)
    try:
        po = api.offt_3d_init(8, 8, 8, custom_params=api.make_params(T1=2, W1=1))
        buf = np.zeros(api.local_elems(po), dtype=np.complex128)
        ptr = buf.ctypes.data_as(C.c_void_p)
        api.lib().offt_3d_execute(po, ptr, ptr, 0)
        assert po.contents.t[api.ALL] >= 99999999.0
        api.offt_3d_fin(po)
    finally:
        cpu_world.uninstall()


def test_baseline_config0_128cube_two_ranks(built, tmp_path):
    """BASELINE.json configs[0]: 128^3 double-complex on 2 ranks, CPU plumbing (no GPU): the product's host
    pipeline over gloo with the reference's default parameters (P1=1, T1=8, W1=2 -- survey_recorded.json)."""
    _run_world(2, [dict(N=[128, 128, 128], params=dict())], tmp_path)
    import json
    meta = json.load(open(tmp_path / "case0_rank0.json"))
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_recorded.json")))
    assert {n: meta["v"][i] for i, n in enumerate(O.PARAM_NAMES)} == rec["default_params_N128_p2"]
