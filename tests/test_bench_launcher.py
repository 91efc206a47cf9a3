"""bench.py as its own multi-rank launcher (no GPU here): `python bench.py --gpus N` without WORLD_SIZE spawns N rank
processes, watches them and relays rank 0's line -- one command, like the reference's `mpiexec ... ./run-fft`
(job-test.sh:9-13, run-fft.c:158-160).  --launch-selftest replaces the GPU work by a gloo rendezvous."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None, timeout=180):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode(), time.time() - t0


def test_two_ranks_are_spawned_and_rank0_line_is_relayed():
    rc, out, err, _ = run(["--gpus", "2", "--launch-selftest"])
    assert rc == 0, err
    line = json.loads(out.strip().splitlines()[-1])
    assert line["selftest"] and line["n_gpus"] == 2 and line["n_ranks_seen"] == 2
    assert line["master"].startswith("127.0.0.1:")


def test_single_rank_needs_no_launcher():
    rc, out, err, _ = run(["--gpus", "1", "--launch-selftest"])
    assert rc == 0, err
    assert json.loads(out.strip().splitlines()[-1])["n_ranks_seen"] == 1


def test_failing_rank_fails_the_run_with_its_tail():
    rc, out, err, dt = run(["--gpus", "2", "--launch-selftest"], env={"OFFT_BENCH_SELFTEST_FAIL_RANK": "1"})
    assert rc != 0
    assert "rank 1 exited with code 3" in err and "fails on purpose" in err
    assert out.strip() == ""      # no half result
    assert dt < 120               # the surviving rank was stopped, nobody waited for a rendezvous time-out


def test_hung_rank_hits_the_wall_clock_limit():
    rc, out, err, dt = run(["--gpus", "2", "--launch-selftest", "--launch-timeout", "6"], env={"OFFT_BENCH_SELFTEST_HANG_RANK": "0"})
    assert rc != 0
    assert "wall-clock limit" in err
    assert dt < 60


def test_under_an_external_launcher_no_ranks_are_spawned():
    """WORLD_SIZE set (torch.distributed.run): the process is a rank, not a launcher"""
    rc, out, err, _ = run(["--gpus", "1", "--launch-selftest"], env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                                                                         MASTER_PORT="29631"))
    assert rc == 0, err
    assert json.loads(out.strip().splitlines()[-1])["n_ranks_seen"] == 1


def test_generic_launcher_for_the_c_harness(tmp_path):
    """tools/launch.py -n N -- <program>: the MPI-free `mpiexec -n N` for the C harness (RANK / LOCAL_RANK /
    WORLD_SIZE / OFFT_ID_FILE in the environment); here with a script that just reports its environment"""
    prog = tmp_path / "who.py"
    prog.write_text("import os, sys\n"
                    "print(os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.path.basename(os.environ['OFFT_ID_FILE']), sys.argv[1])\n"
                    "sys.exit(4 if os.environ.get('FAIL_RANK') == os.environ['RANK'] else 0)\n")
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch.py"), "-n", "3", "--", str(prog), "hello"], env=e,
                       capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    assert p.stdout.decode().split() == ["0", "3", "0", "rccl_id", "hello"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch.py"), "-n", "3", "--", str(prog), "x"], env=dict(e, FAIL_RANK="2"),
                       capture_output=True, timeout=120)
    assert p.returncode != 0 and "rank 2 exited with code 4" in p.stderr.decode()


def test_line_blocks_are_self_consistent_across_n():
    """the derived blocks of the JSON line, from synthetic measurements (pure functions, no GPU): the y/x pair is reported as
    a pair, recorded PMC traffic is labelled as recorded, and a group of one has no xGMI rate"""
    sys.path.insert(0, ROOT)
    import bench
    E, esz = 1024.0 ** 3, 16
    # N = 1, y and x launches alternating: only z and the pair's sum exist
    r = bench.roofline_block([5.8e-3, 5.3e-3, 5.3e-3], 6, False, esz, E, 1, 16.4e-3, 16.4e-3, lambda axis: 34.365e9 if axis == "z" else None)
    assert set(r["pass_ms"]) == {"z", "y+x (alternating launches, measured as a pair)"}
    assert abs(r["pass_ms"]["y+x (alternating launches, measured as a pair)"] - 10.6) < 1e-9
    assert r["kernel"].startswith("fft_panel_k (z-axis pass)") and abs(r["frac"] - 2 * esz * E / 5.8e-3 / 8e12) < 1e-3
    assert r["traffic"] == 34.365e9 and "NOT measured in this run" in r["traffic_source"]
    assert r["frac"] <= 1.0 and r["avg_launch_ms"] <= 16.4
    # N = 1, three plain launches: three separate times, the slowest is named
    r = bench.roofline_block([5.0e-3, 6.0e-3, 5.5e-3], 0, False, esz, E, 1, 16.6e-3, 16.5e-3)
    assert set(r["pass_ms"]) == {"z", "y", "x"} and "y-axis" in r["kernel"] and r["traffic"] is None and r["traffic_source"] is None
    # N = 8: K1 phase and the chunked phase, per-rank bytes
    r = bench.roofline_block([5.4e-3, 6.0e-3, 5.4e-3], 5, False, esz, E, 1, 16.9e-3, 16.8e-3)   # x-y-z layout: z and x alternate
    assert set(r["pass_ms"]) == {"y", "z+x (alternating launches, measured as a pair)"} and "y-axis" in r["kernel"]
    r = bench.roofline_block([0.8e-3, 0.0, 4.0e-3], 0, True, esz, E, 8, 4.9e-3, 4.8e-3)
    assert "K1" in r["kernel"] and r["alg_bytes_per_launch"] == 2 * esz * E / 8 and len(r["pass_ms"]) == 2
    # xGMI block: 8 ranks on 7 links; a group of one (the one-GPU rehearsal) has none
    x = bench.xgmi_block(8, esz, E, 8, 4.0e-3)
    assert x["links_used"] == 7 and abs(x["bytes_per_link_per_direction"] - esz * E / 64) < 1
    assert bench.xgmi_block(1, esz, E, 1, 14e-3) is None
    # the rank code times the SAME mode at every N and adds the pipelined figure for N > 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "enqueue_only=False)" in src and '"ms_per_step_pipelined"' in src and '"ms_per_step_sync"' in src
