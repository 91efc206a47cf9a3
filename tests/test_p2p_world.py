"""The DIRECT-STORE exchange (OFFT_EXCHANGE=p2p / offt_hip_set_exchange) on the CPU: the product's host logic -- peer
mappings, per-block base tables into the peers' volumes, READY / FREE flags, the barrier-mirrored inverse -- with several
ranks as threads of one process on the test-only CPU descriptor interpreter (tests/_thread_world.py ... cpu).  A pass of
one rank really writes into another rank's receive volume; there is no transport for the data.  The same cases run on
the GPU in tests/test_gpu_world.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cpu_thread_world(size, cases, tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_thread_world.py"), str(size), json.dumps(cases), str(tmp_path), "cpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1", OFFT_P2P_TIMEOUT="20"))
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    summary = json.load(open(tmp_path / "summary.json"))
    assert len(summary) == len(cases)
    for rec in summary:
        for k in ("rel_numpy", "rel_oracle", "rel_inverse"):
            if k in rec:
                assert rec[k] < rec["tol"], rec
    return summary


def test_staged_exchange_in_a_thread_world(built, tmp_path):
    """the thread world itself (staged exchange through the in-process wire), slab and pencil"""
    run_cpu_thread_world(2, [dict(N=[8, 8, 8], params=dict(P1=1)), dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=2), inv=2)], tmp_path)


def test_direct_store_two_ranks(built, tmp_path):
    cases = [dict(N=[8, 8, 8], params=dict(P1=1), p2p=1), dict(N=[8, 8, 8], params=dict(P1=2), p2p=1),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=2, T2=2), p2p=1, repeat=2),
             dict(N=[10, 6, 9], params=dict(P1=1, T1=3, T2=2), p2p=1, inv=2, repeat=1),          # ragged tiles and chunks
             dict(N=[9, 7, 11], params=dict(P1=2, T1=2, T2=3), p2p=1, inv=2, repeat=2),          # pencil, uneven blocks
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, W1=1, T2=2), p2p=1, repeat=3),            # ring of 2 slots reused
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, W1=0), p2p=1, repeat=2),                  # ring of 1
             dict(N=[8, 8, 8], params=dict(P1=1, S=1, T1=2, W1=1), p2p=1, inv=1),                # x-y-z layout: pencil schedule on 1 x p
             dict(N=[8, 8, 8], params=dict(P1=2), eq=1, p2p=1), dict(N=[8, 6, 12], params=dict(P1=2, S=1), r2c=1, p2p=1),
             dict(N=[8, 8, 8], params=dict(P1=1), r2c=1, p2p=1),
             dict(N=[8, 8, 8], params=dict(P1=2, T1=2, T2=2), p2p=1, env=dict(OFFT_PENCIL_ZC_LAYOUT=1)),
             dict(N=[8, 8, 8], params=dict(P1=1, T1=2, T2=2), p2p=1, env=dict(OFFT_SLAB_XC_LAYOUT=1)),
             dict(N=[16, 8, 8], params=dict(P1=1), f32=1, p2p=1, inv=1)]
    run_cpu_thread_world(2, cases, tmp_path)


def test_direct_store_three_and_four_ranks(built, tmp_path):
    run_cpu_thread_world(3, [dict(N=[8, 10, 9], params=dict(P1=1), p2p=1, repeat=1), dict(N=[9, 9, 7], params=dict(P1=3), p2p=1, inv=1),
                             dict(N=[10, 7, 9], params=dict(P1=1, S=1), p2p=1)], tmp_path)
    run_cpu_thread_world(4, [dict(N=[8, 8, 8], params=dict(P1=2), p2p=1, repeat=2), dict(N=[10, 6, 9], params=dict(P1=2, T1=2, W1=1, T2=2), p2p=1, inv=1, repeat=1),
                             dict(N=[8, 8, 8], params=dict(P1=4, S=1), p2p=1), dict(N=[8, 8, 8], params=dict(P1=1, T1=3), p2p=1, inv=1),
                             dict(N=[12, 8, 10], params=dict(P1=4, T1=1, T2=3), p2p=1)], tmp_path)


def test_direct_store_eight_ranks_bench_meshes(built, tmp_path):
    """the meshes of the multi-GPU bench: 1 x 8 (headline), the reference default 2 x 4, 8 x 1; ragged; inverse"""
    cases = [dict(N=[16, 16, 16], params=dict(P1=1), p2p=1, repeat=2), dict(N=[16, 16, 16], params=dict(), p2p=1, repeat=2),
             dict(N=[16, 16, 16], params=dict(P1=8), p2p=1), dict(N=[20, 12, 18], params=dict(P1=2, T1=3, W1=1, T2=2), p2p=1, inv=1),
             dict(N=[16, 16, 16], params=dict(P1=4, T2=2), p2p=1, inv=1, repeat=1), dict(N=[16, 16, 16], params=dict(P1=1), p2p=1, inv=1, repeat=1)]
    s = run_cpu_thread_world(8, cases, tmp_path)
    assert s[1]["mesh"] == [2, 4]
