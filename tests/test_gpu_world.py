"""-m gpu: the multi-rank schedules (slab and pencil) with 2 and 4 ranks SHARING the one GPU of the test box.
Every kernel launch, descriptor, device buffer, stream and event is the product's; only the transport is
swapped (RCCL refuses two ranks on one device) for a host-staged gloo exchange.  Results are gathered
through ostart/osize/ostride and compared with numpy.fft and the oracle on the same decomposition."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cpu_world
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(size, cases, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(size):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gpu_world_worker.py"),
                                       json.dumps(cases), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    for ci, case in enumerate(cases):
        shape = tuple(case["N"])
        r2c = case.get("r2c", 0)
        oshape = (shape[0], shape[1], shape[2] // 2 + 1) if r2c else shape
        G = np.full(oshape, np.nan + 0j)
        for r in range(size):
            meta = json.load(open(tmp_path / f"case{ci}_rank{r}.json"))
            cpu_world.scatter_out(meta["comm"], np.load(tmp_path / f"case{ci}_rank{r}.npy"), G)
        assert not np.isnan(G).any(), case
        want = np.fft.rfftn(O.hash_field(*shape).real, axes=(0, 1, 2)) if r2c else np.fft.fftn(O.hash_field(*shape))
        assert np.linalg.norm(G - want) / np.linalg.norm(want) < 1e-13, case
        og, _, _ = O.world_fft(*shape, size, kind=1, is_equalxy=case.get("eq", 0), is_r2c=r2c, **case["params"])
        assert np.linalg.norm(G - og) / np.linalg.norm(og) < 1e-13, case
        if case.get("inv"):
            for r in range(size):
                meta = json.load(open(tmp_path / f"case{ci}_rank{r}.json"))["comm"]
                back = np.load(tmp_path / f"case{ci}_rank{r}_inv.npy")
                blk = O.hash_field(*meta["isize"], *meta["istart"])
                assert np.linalg.norm(back / np.prod(shape) - blk) / np.linalg.norm(blk) < 1e-13, (case, r)
        if case.get("p2p"):  # the direct-store exchange was really in use (no silent fall-back to the staged one)
            assert all(json.load(open(tmp_path / f"case{ci}_rank{r}.json"))["exchange"] == 1 for r in range(size)), case


def run_thread_world(size, cases, tmp_path, env=None):
    """`size` ranks as threads of ONE process on the one GPU (the box admits at most 6 processes on the card).
    GPU_MAX_HW_QUEUES: HIP maps a process's streams onto a few hardware queues (4 by default) and two streams sharing one run
    in order -- a missing event edge between them would then go unnoticed, depending on which streams happened to land
    together (tools/async_negative_control.sh caught edges 2 and 5 in one run and not in the next until the queues were
    raised above the number of streams)."""
    env = dict({"GPU_MAX_HW_QUEUES": "24"}, **(env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_thread_world.py"), str(size), json.dumps(cases), str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    summary = json.load(open(tmp_path / "summary.json"))
    assert len(summary) == len(cases)
    for rec in summary:
        for k in ("rel_numpy", "rel_oracle", "rel_inverse"):
            if k in rec:
                assert rec[k] < rec["tol"], rec
    return summary


def test_eight_ranks_one_gpu_bench_meshes(built, tmp_path):
    """BASELINE configs[3] as far as one GPU can take it: the 8-rank meshes of the multi-GPU bench -- 1x8 (headline),
    the reference default 2x4, 8x1 -- at 256^3 and on a ragged grid, forward and inverse, r2c"""
    cases = [dict(N=[256, 256, 256], params=dict(P1=1)), dict(N=[256, 256, 256], params=dict()),
             dict(N=[256, 256, 256], params=dict(P1=8)), dict(N=[256, 256, 256], params=dict(P1=2), env=dict(OFFT_MIN_MSG=0)),
             dict(N=[100, 72, 90], params=dict(P1=2, T1=7, T2=5)), dict(N=[100, 72, 90], params=dict(P1=1)),
             dict(N=[128, 128, 128], params=dict(), inv=1), dict(N=[128, 128, 128], params=dict(P1=1), inv=1),
             dict(N=[128, 64, 256], params=dict(P1=4), r2c=1), dict(N=[128, 128, 128], params=dict(P1=4, S=1, T2=8)),
             # the first-version exchange-volume layouts (kept for uneven grids) on an even grid
             dict(N=[128, 128, 128], params=dict(P1=2, T1=16, T2=8), env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[128, 128, 128], params=dict(P1=1, T1=32, T2=4), env=dict(OFFT_SLAB_XC_LAYOUT=1)),
             dict(N=[128, 128, 128], params=dict(P1=2, T1=16, T2=8), inv=1),
             # row and column exchanges on two comm streams (OFFT_COMM_STREAMS=2): other event edges, same result
             dict(N=[128, 128, 128], params=dict(P1=2, T1=16, T2=8), env=dict(OFFT_COMM_STREAMS=2)),
             dict(N=[128, 128, 128], params=dict(P1=4, T1=8, T2=16, W2=0), env=dict(OFFT_COMM_STREAMS=2))]
    s = run_thread_world(8, cases, tmp_path)
    assert s[1]["mesh"] == [2, 4]  # offt-compute.c:3138-3139: the largest divisor of p that is <= sqrt(p)


def test_config3_full_size_eight_ranks_one_gpu(built, tmp_path):
    """BASELINE configs[3] at FULL size on one GPU: 1024^3 double-complex over 8 ranks (threads sharing the card, 8 GiB of
    buffers each), the bench's 1x8 mesh and the reference-default 2x4 pencil mesh.  Checked through size-independent
    properties: Parseval over all ranks, and the closed form of the harness ramp at spots on the three axes, off the axes
    and in the last rank's block (cube side n: X[0,0,0] = n^3 111 (n-1)/2, X[0,0,k] = n^3 (-1/2 + i/2 cot(pi k/n)))"""
    n = 1024
    spots = [[0, 0, 0], [0, 0, 1], [0, 0, 3], [0, 0, 1023], [0, 5, 0], [0, 1000, 0], [7, 0, 0], [513, 0, 0], [1, 1, 1], [1023, 1023, 1023],
             [0, 511, 640], [300, 0, 900]]
    cases = [dict(N=[n, n, n], params=dict(P1=1), check="ramp", spots=spots), dict(N=[n, n, n], params=dict(), check="ramp", spots=spots)]
    s = run_thread_world(8, cases, tmp_path)
    assert s[0]["mesh"] == [1, 8] and s[1]["mesh"] == [2, 4]


def test_single_precision_worlds_one_gpu(built, tmp_path):
    """BASELINE configs[4] as far as one GPU can take it: single-precision worlds of 2 and 8 ranks (slab, default
    pencil mesh, uneven per-peer blocks), tolerance rel-L2 <= 5e-6"""
    cases2 = [dict(N=[128, 128, 128], params=dict(P1=1), f32=1), dict(N=[128, 128, 128], params=dict(P1=2), f32=1),
              dict(N=[64, 64, 2048], params=dict(P1=1), f32=1), dict(N=[2048, 32, 32], params=dict(P1=2), f32=1, inv=1)]
    run_thread_world(2, cases2, tmp_path)
    cases8 = [dict(N=[256, 256, 256], params=dict(P1=1), f32=1), dict(N=[256, 256, 256], params=dict(), f32=1),
              dict(N=[64, 2048, 64], params=dict(P1=8), f32=1), dict(N=[128, 128, 128], params=dict(), f32=1, inv=1)]
    run_thread_world(8, cases8, tmp_path)
    cases3 = [dict(N=[64, 64, 64], params=dict(P1=1), f32=1), dict(N=[128, 100, 96], params=dict(P1=3), f32=1)]
    run_thread_world(3, cases3, tmp_path)  # uneven F / F+1 blocks in single precision


def test_two_ranks_one_gpu(built, tmp_path):
    cases = [dict(N=[64, 64, 64], params=dict(P1=1)), dict(N=[64, 64, 64], params=dict(P1=2)),
             dict(N=[128, 64, 32], params=dict(P1=1, T1=16, T2=4)), dict(N=[64, 64, 64], params=dict(P1=1, S=1)),
             dict(N=[64, 64, 64], params=dict(P1=2), eq=1), dict(N=[18, 20, 14], params=dict(P1=1, T1=4, T2=3)),
             dict(N=[64, 32, 128], params=dict(P1=1), r2c=1), dict(N=[256, 256, 256], params=dict(P1=1)),
             dict(N=[64, 64, 64], params=dict(P1=1), inv=1), dict(N=[64, 32, 16], params=dict(P1=2, T1=8), inv=1),
             # mixed-radix panel kernels with per-peer splits that are not powers of two (96 = 2 x 48, 120 = 2 x 60)
             dict(N=[96, 96, 96], params=dict(P1=1)), dict(N=[120, 96, 100], params=dict(P1=2), inv=1),
             dict(N=[96, 120, 96], params=dict(P1=1), r2c=1),
             # lengths with the prime factor 127: Bluestein passes addressing per-peer blocks (254 = 2 x 127 per peer)
             dict(N=[254, 64, 508], params=dict(P1=1)), dict(N=[64, 254, 508], params=dict(P1=2), inv=1)]
    run_world(2, cases, tmp_path)


def test_four_ranks_one_gpu(built, tmp_path):
    cases = [dict(N=[64, 64, 64], params=dict(P1=1)), dict(N=[64, 64, 64], params=dict(P1=2)),
             dict(N=[64, 64, 64], params=dict(P1=4, S=1)), dict(N=[128, 128, 128], params=dict(P1=1)),
             dict(N=[20, 12, 18], params=dict(P1=2, T1=3, W1=1)), dict(N=[128, 128, 128], params=dict()),
             dict(N=[96, 96, 96], params=dict(P1=1)), dict(N=[192, 96, 120], params=dict(P1=2))]
    run_world(4, cases, tmp_path)


def test_three_ranks_one_gpu_uneven_blocks(built, tmp_path):
    """3 ranks: nothing divides evenly, every pass addresses the reference's F / F+1 per-peer blocks
    (offt-compute.c:132-144) -- on the register kernels (fft_panelx_k instances), not the any-length kernel"""
    cases = [dict(N=[64, 64, 64], params=dict(P1=1)), dict(N=[128, 64, 256], params=dict(P1=1, T1=8)),
             dict(N=[100, 96, 120], params=dict(P1=3)), dict(N=[64, 100, 128], params=dict(P1=1), inv=1),
             dict(N=[128, 96, 64], params=dict(P1=1), r2c=1), dict(N=[20, 14, 18], params=dict(P1=1, S=1))]
    run_world(3, cases, tmp_path)



def test_direct_store_exchange_thread_worlds(built, tmp_path):
    """The direct-store exchange (OFFT_EXCHANGE=p2p) on the GPU, ranks as threads sharing the card: the packing kernels of
    one rank store straight into the other ranks' receive volumes (a peer pointer here is another thread's device buffer),
    flag kernels order the streams -- no transport, no host synchronisation inside a transform.  The bench meshes 1 x 8,
    2 x 4, 8 x 1, ragged grids, single precision, the barrier-mirrored inverse, and plans used several times in a row
    (buffer reuse: FREE flags); forward -> inverse -> forward on one plan."""
    cases = [dict(N=[256, 256, 256], params=dict(P1=1), p2p=1, repeat=2), dict(N=[256, 256, 256], params=dict(), p2p=1, repeat=2),
             dict(N=[256, 256, 256], params=dict(P1=8), p2p=1, repeat=1),
             dict(N=[100, 72, 90], params=dict(P1=2, T1=7, T2=5), p2p=1, repeat=1), dict(N=[100, 72, 90], params=dict(P1=1), p2p=1, inv=1),
             dict(N=[128, 128, 128], params=dict(), p2p=1, inv=1, repeat=1), dict(N=[128, 128, 128], params=dict(P1=1), p2p=1, inv=1, repeat=1),
             dict(N=[128, 64, 256], params=dict(P1=4), r2c=1, p2p=1), dict(N=[128, 128, 128], params=dict(P1=4, S=1, T2=8), p2p=1),
             dict(N=[128, 128, 128], params=dict(P1=2, T1=16, T2=8), p2p=1, env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[128, 128, 128], params=dict(P1=1, T1=32, T2=4), p2p=1, env=dict(OFFT_SLAB_XC_LAYOUT=1)),
             dict(N=[256, 256, 256], params=dict(P1=1), f32=1, p2p=1, repeat=1), dict(N=[256, 256, 256], params=dict(), f32=1, p2p=1, inv=1)]
    s = run_thread_world(8, cases, tmp_path)
    assert s[1]["mesh"] == [2, 4]
    run_thread_world(3, [dict(N=[64, 64, 64], params=dict(P1=1), p2p=1, repeat=1), dict(N=[100, 96, 120], params=dict(P1=3), p2p=1, inv=1),
                         dict(N=[128, 100, 96], params=dict(P1=3), f32=1, p2p=1)], tmp_path)
    # ... and with SLOW passes (the test build holds the stream 50 ms ahead of every pass): the host runs far ahead of the
    # device, every READY / FREE wait really has to hold its stream -- plans reused, so that a rank's next transform stores
    # into its peer's volume only after the peer has consumed the previous one
    run_thread_world(2, [dict(N=[128, 128, 128], params=dict(P1=1, T1=32, T2=32), p2p=1, repeat=1),
                         dict(N=[128, 128, 128], params=dict(P1=2, T1=32, W1=0, T2=32), p2p=1, inv=1)], tmp_path,
                     env=dict(OFFT_TEST_SLOW_PASS_MS="50", OFFT_P2P_TIMEOUT="60"))
    # ... and with only the ODD rank slow: the even rank runs a whole phase ahead of it.  In this world a dropped READY wait and
    # a dropped FREE wait both come out as wrong results (tools/async_negative_control.sh, profiles/r03_async_negative_control.txt)
    run_thread_world(2, [dict(N=[128, 128, 128], params=dict(P1=1, T1=32, T2=16), p2p=1, repeat=2),
                         dict(N=[128, 128, 128], params=dict(P1=2, T1=16, W1=1, T2=16), p2p=1, repeat=2)], tmp_path,
                     env=dict(OFFT_TEST_SLOW_PASS_MS="50", OFFT_TEST_SLOW_RANKS="odd", OFFT_P2P_TIMEOUT="60"))


def test_direct_store_exchange_full_size_eight_ranks_one_gpu(built, tmp_path):
    """BASELINE configs[3] at full size with the direct-store exchange: 1024^3 over 8 thread-ranks, 1 x 8 and 2 x 4"""
    n = 1024
    spots = [[0, 0, 0], [0, 0, 1], [0, 0, 1023], [0, 5, 0], [7, 0, 0], [1, 1, 1], [1023, 1023, 1023], [0, 511, 640], [300, 0, 900]]
    cases = [dict(N=[n, n, n], params=dict(P1=1), check="ramp", spots=spots, p2p=1), dict(N=[n, n, n], params=dict(), check="ramp", spots=spots, p2p=1)]
    s = run_thread_world(8, cases, tmp_path)
    assert s[0]["mesh"] == [1, 8] and s[1]["mesh"] == [2, 4]


def test_direct_store_exchange_between_processes_hipipc(built, tmp_path):
    """... and between PROCESSES sharing the card: the product's hipIpc path (hipIpcGetMemHandle, handles gathered with the
    exchange primitive, hipIpcOpenMemHandle) -- what a one-process-per-GPU run uses, here with both ends on one device"""
    cases = [dict(N=[64, 64, 64], params=dict(P1=1), p2p=1, repeat=2), dict(N=[64, 64, 64], params=dict(P1=2), p2p=1, repeat=1),
             dict(N=[128, 64, 32], params=dict(P1=1, T1=16, T2=4), p2p=1, inv=1), dict(N=[18, 20, 14], params=dict(P1=1, T1=4, T2=3), p2p=1),
             dict(N=[64, 32, 16], params=dict(P1=2, T1=8), p2p=1, inv=1, repeat=1)]
    run_world(2, cases, tmp_path)
    run_world(4, [dict(N=[64, 64, 64], params=dict(P1=2), p2p=1, repeat=1), dict(N=[128, 128, 128], params=dict(P1=1), p2p=1)], tmp_path)


def test_long_lines_in_worlds(built, tmp_path):
    """lines beyond the single-launch kernels on three ranks: a long prime (Bluestein through scratch) on the z axis of the
    slab schedule and on the y axis of a 3 x 1 pencil mesh -- the per-peer blocks of 10007 points are uneven --, a four-step
    length whose uneven per-peer blocks the decomposition cannot follow (6250 over 3: through scratch as well), a long
    real-input line; forward against numpy and the oracle, mirrored inverse"""
    run_thread_world(3, [dict(N=[4, 6, 10007], params=dict(P1=1), inv=1), dict(N=[6, 10007, 4], params=dict(P1=3)),
                         dict(N=[4, 6, 6250], params=dict(P1=1), inv=1), dict(N=[6, 4, 12000], params=dict(P1=1), r2c=1),
                         dict(N=[6, 4, 2038], params=dict(P1=1), r2c=1), dict(N=[6, 6, 1016], params=dict(P1=3), r2c=1)], tmp_path)  # real input along Bluestein lengths


def test_staged_exchange_with_an_asynchronous_transport(built, tmp_path):
    """The staged (default) exchange with NOTHING drained on the host: the test transport enqueues device-to-device copies
    on the ranks' comm streams, ordered between ranks by events only (tests/_thread_world.py, "async"), so the schedules'
    own event edges between compute and comm streams are all that orders kernels and exchanges -- as with RCCL.  A missing
    edge (a K2 that does not wait for its tile, a K1 that overwrites a send block still being read, the next transform
    running into the previous one's exchange) shows up as a wrong result.  Slab and pencil meshes of the bench, ragged
    grids, two comm streams, the mirrored inverse, plans used several times in a row."""
    cases = [dict(N=[256, 256, 256], params=dict(P1=1), repeat=2, **{"async": 1}), dict(N=[256, 256, 256], params=dict(), repeat=2, **{"async": 1}),
             dict(N=[128, 128, 128], params=dict(P1=8), repeat=1, **{"async": 1}),
             dict(N=[100, 72, 90], params=dict(P1=2, T1=7, T2=5), repeat=1, **{"async": 1}),
             dict(N=[128, 128, 128], params=dict(), inv=1, repeat=1, **{"async": 1}), dict(N=[128, 128, 128], params=dict(P1=1), inv=1, repeat=1, **{"async": 1}),
             dict(N=[128, 128, 128], params=dict(P1=2, T1=16, T2=8), repeat=1, env=dict(OFFT_COMM_STREAMS=2), **{"async": 1}),
             dict(N=[128, 128, 128], params=dict(P1=1, T1=16, T2=4), repeat=1, env=dict(OFFT_SELF_BYPASS=0), **{"async": 1}),
             dict(N=[128, 128, 128], params=dict(P1=1), f32=1, repeat=1, **{"async": 1})]
    run_thread_world(8, cases, tmp_path)
    run_thread_world(3, [dict(N=[64, 100, 128], params=dict(P1=1), inv=1, repeat=1, **{"async": 1}), dict(N=[100, 96, 120], params=dict(P1=3), repeat=1, **{"async": 1})], tmp_path)
    # two ranks: the host threads get in each other's way least, so the device really runs behind the host -- this is the
    # world in which tools/async_negative_control.sh sees a dropped edge ("K2 waits for its chunk") as a wrong result
    # (profiles/r03_async_negative_control.txt; in an 8-thread world the slow host rendezvous hides it)
    run_thread_world(2, [dict(N=[256, 256, 256], params=dict(P1=1), repeat=2, **{"async": 1}), dict(N=[256, 256, 256], params=dict(P1=2), repeat=2, **{"async": 1}),
                         dict(N=[256, 256, 256], params=dict(P1=1, T1=32, T2=8), inv=2, repeat=1, **{"async": 1}),
                         dict(N=[256, 128, 256], params=dict(P1=2, T1=16, W1=1, T2=16), inv=2, repeat=1, **{"async": 1}),
                         dict(N=[256, 256, 256], params=dict(P1=1, S=1, T1=16, W1=1, T2=16), inv=1, repeat=1, **{"async": 1}),
                         dict(N=[256, 256, 256], params=dict(P1=1, S=1), repeat=1, **{"async": 1}), dict(N=[256, 256, 256], params=dict(P1=1), f32=1, repeat=1, **{"async": 1})], tmp_path)
    # the other half: a FAST wire ("async": 2) and SLOW passes (the test build holds the stream 50 ms ahead of every pass), so
    # that the host runs far ahead of the device and an exchange that did not wait for the kernel packing its data would
    # read too early.  All numbered edges of offt_host.c (forward and mirrored schedules), dropped one at a time, come out wrong in these worlds:
    # profiles/r03_async_negative_control.txt
    run_thread_world(2, [dict(N=[256, 256, 256], params=dict(P1=1, T1=64, T2=32), repeat=1, **{"async": 2}),
                         dict(N=[256, 256, 256], params=dict(P1=1, T1=64, T2=32), inv=1, **{"async": 2}),
                         dict(N=[256, 256, 256], params=dict(P1=2, T1=32, W1=1, T2=32), repeat=1, **{"async": 2}),
                         dict(N=[256, 256, 256], params=dict(P1=2, T1=32, W1=1, T2=32), inv=1, **{"async": 2}),
                         dict(N=[256, 256, 256], params=dict(P1=1, S=1, T1=32, W1=1, T2=32), inv=1, **{"async": 2})], tmp_path,
                     env=dict(OFFT_TEST_SLOW_PASS_MS="50"))
