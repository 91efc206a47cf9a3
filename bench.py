#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native OFFT hot path.

Metric (BASELINE.json): 3-D FFT GFLOP/s + fraction of the HBM roofline, 1024^3
double-complex, forward transform (offt_3d_execute), data resident in HBM.

  python bench.py --gpus N --steps K --warmup W [--dtype f64|f32] [--n 1024]

One process per GPU.  Either the caller starts the ranks (torch.distributed.run sets RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_*), or -- when WORLD_SIZE is unset and N > 1 -- this script is the
launcher, like the reference's one-command `mpiexec ... ./run-fft` (job-test.sh:9-13): the parent
spawns N rank processes BEFORE touching the GPU (it never imports torch), watches them against a
wall-clock limit, kills the others when one fails, relays rank 0's line and exits non-zero with the
failing rank's tail otherwise.  Ranks synchronise themselves (barrier before and after the timed
steps, max over ranks; run-fft.c:309, 371-414).

A "step" is one synchronous offt_3d_execute of the whole grid, at every N (the reference harness
times one call per repetition between barriers, run-fft.c:374-395).  flops = 5 E log2 E,
algorithmic bytes = 6 * S * E / P per GPU per transform (S = 16 B double-complex, 8 B
single-complex).  Rank 0 prints ONE JSON line.  For N > 1 the headline is the 1 x N
mesh (one exchange over all N - 1 xGMI links) in that same synchronous mode; the line also carries
the pipelined figure (`ms_per_step_pipelined`: the K steps enqueued back to back, one wait), the
reference-default mesh (2 x 4 at N = 8), an exchange-only and a compute-only time of the headline
schedule, the per-link rate against 153 GB/s, an xGMI link probe and the number of ranks that
answered an all-reduce.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable copy)
XGMI_LINK = 153e9   # B/s per link (7 links per GPU)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1024, help="grid side (default 1024: BASELINE configs[2])")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="f64: double-complex (headline); f32: single-complex (configs[4])")
    ap.add_argument("--p1", type=int, default=-1, help="mesh rows of the headline; default 1 (one exchange over all xGMI links)")
    ap.add_argument("--layout", default="zyx", choices=["zyx", "xyz"], help="output layout: reference default z-y-x, or S=1 x-y-z")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="grid side of the CPU baseline (0: 1024 if the host has >= 64 GiB free, else 512)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: headline only (no pencil mesh, split timings, link probe)")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="wall-clock limit of the whole run in seconds")
    ap.add_argument("--launch-selftest", action="store_true", help="rehearse the launcher on CPU: ranks rendezvous over gloo, no GPU work")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher (parent): no torch, no HIP
# ------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv, script=None, interpreter=sys.executable):
    import signal
    import tempfile
    n = args.gpus
    port = _free_port()
    logdir = tempfile.mkdtemp(prefix="offt_bench_")
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OFFT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        out = open(os.path.join(logdir, f"rank{r}.out"), "wb")
        err = open(os.path.join(logdir, f"rank{r}.err"), "wb")
        logs.append((out, err))
        procs.append(subprocess.Popen(([interpreter] if interpreter else []) + [os.path.abspath(script or __file__)] + argv, env=env, stdout=out, stderr=err,
                                      start_new_session=True))

    def tail(r, nbytes=3000):
        txt = ""
        for kind in ("out", "err"):
            try:
                with open(os.path.join(logdir, f"rank{r}.{kind}"), "rb") as f:
                    f.seek(0, 2)
                    size = f.tell()
                    f.seek(max(0, size - nbytes))
                    txt += f"--- rank {r} std{kind} (tail) ---\n" + f.read().decode(errors="replace") + "\n"
            except OSError:
                pass
        return txt

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)  # the child's own session: only processes this launcher started
                except OSError:
                    pass
        t_end = time.time() + 10
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass

    t0 = time.time()
    failed, why = None, ""
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed, why = bad[0], f"rank {bad[0]} exited with code {codes[bad[0]]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > args.launch_timeout:
                running = [r for r, c in enumerate(codes) if c is None]
                failed, why = running[0], f"wall-clock limit of {args.launch_timeout:.0f} s reached; ranks still running: {running}"
                break
            time.sleep(0.05)
    finally:
        kill_all()
        for out, err in logs:
            out.close()
            err.close()
    if failed is not None:
        sys.stderr.write(f"bench.py launcher: {why}\n{tail(failed)}")
        if failed != 0:
            sys.stderr.write(tail(0, 1500))
        return 1
    with open(os.path.join(logdir, "rank0.out"), "rb") as f:
        sys.stdout.write(f.read().decode(errors="replace"))
    sys.stdout.flush()
    return 0


def selftest_rank(args):
    """launcher rehearsal on CPU: env plumbing + gloo rendezvous + an all-reduce of ones; optional injected faults"""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("OFFT_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        print(f"selftest: rank {rank} fails on purpose", file=sys.stderr, flush=True)
        sys.exit(3)
    if os.environ.get("OFFT_BENCH_SELFTEST_HANG_RANK") == str(rank):
        time.sleep(3600)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.int32)
    dist.all_reduce(ones)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": args.gpus, "n_ranks_seen": int(ones.item()),
                          "local_ranks": "LOCAL_RANK == RANK", "master": os.environ["MASTER_ADDR"] + ":" + os.environ["MASTER_PORT"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(n, cores, reps):
    """The oracle (CPU restatement of the reference pipeline: default parameters, tiles, pack ->
    all-to-all -> unpack, 1-D FFTs) timed on this host: one simulated rank per core as OpenMP
    threads.  Checker infrastructure used here ONLY as the reported CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    L = O.lib()
    p = 1
    while p * 2 <= cores:
        p *= 2
    w = L.orc_world_create(n, n, n, p, 0, 0, 0, None)
    L.orc_world_fill(w, 1)
    L.orc_world_execute(w, p)  # warm-up: first touch of the exchange buffers (the GPU side is warmed up too)
    dt = None
    for _ in range(reps):      # min of reps, like the reference harness (run-fft.c:408-413)
        L.orc_world_fill(w, 1)
        t0 = time.time()
        L.orc_world_execute(w, p)
        d = time.time() - t0
        dt = d if dt is None else min(dt, d)
    L.orc_world_destroy(w)
    flops = 5.0 * n ** 3 * math.log2(n ** 3)
    return {"value": round(flops / dt / 1e9, 3), "unit": "GFLOP/s", "cores": p, "kind": "port",
            "sample": f"{n}^3 double-complex forward, best of {reps} after 1 warm-up, {p} simulated MPI ranks (one OpenMP thread each), "
                      f"reference default parameters, {dt:.2f} s per transform"}


def mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


# ------------------------------------------------------------------------------------------------
# the JSON line's derived blocks: pure functions of the measurements (tests/test_bench_launcher.py checks the keys on CPU)
# ------------------------------------------------------------------------------------------------
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def roofline_block(pass_s, paired, multi, esz, E, world, dt_step, dev_s, traffic_lookup=None):
    """roofline of the dominant kernel.  pass_s = device seconds of the z, y, x passes of one transform (events on the
    plan's stream); `paired` = 0, or the two passes (bits: 1 = z, 2 = y, 4 = x; offt_hip_last_passes_paired) whose launches
    alternate over groups of planes, so that only their SUM was measured."""
    names = ["z", "y", "x"]
    alg_launch = 2.0 * esz * E / world
    if multi:
        # multi-rank slab schedule: phase 0 = the FFTz launches (K1, all x-tiles, no waiting on the exchange);
        # phase 2 = z-chunks of exchange-wait + FFTy + FFTx, which includes time on the wire
        k, kdur, kname = 0, pass_s[0], "fft_panel_k (z-axis pass K1, all x-tiles)"
        pass_ms = {"K1 (FFTz + pack, all x-tiles)": round(pass_s[0] * 1e3, 4),
                   "exchange-wait + K2 + K3 (z-chunks)": round(pass_s[2] * 1e3, 4)}
    elif paired:
        # the pair's launches interleave: each of the two moves 2*S*E bytes, the slower one is not known separately, so the
        # pair enters with HALF its time per launch-equivalent -- the third pass, a single launch, is named when it is slower
        ab = [i for i in range(3) if paired >> i & 1]
        one = [i for i in range(3) if not paired >> i & 1][0]
        pair = pass_s[ab[0]] + pass_s[ab[1]]
        if pass_s[one] >= 0.5 * pair:
            k, kdur, kname = one, pass_s[one], f"fft_panel_k ({names[one]}-axis pass)"
        else:
            k, kdur, kname = ab[0], 0.5 * pair, f"fft_panel_k ({names[ab[0]]}- and {names[ab[1]]}-axis passes, alternating launches: half of the pair's time)"
        pass_ms = {names[one]: round(pass_s[one] * 1e3, 4),
                   f"{names[ab[0]]}+{names[ab[1]]} (alternating launches, measured as a pair)": round(pair * 1e3, 4)}
    else:
        k = max(range(3), key=lambda i: pass_s[i])
        kdur, kname = pass_s[k], f"fft_panel_k ({names[k]}-axis pass)"
        pass_ms = {names[i]: round(pass_s[i] * 1e3, 4) for i in range(3)}
    traffic = traffic_lookup(names[k]) if traffic_lookup else None
    roof = {"bound": "hbm", "kernel": kname, "achieved": round(alg_launch / kdur / 1e9, 1) if kdur > 0 else None,
            "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(alg_launch / kdur / HBM_PEAK, 4) if kdur > 0 else None,
            "traffic": traffic,
            "traffic_source": ("profiles/pmc_traffic.json (recorded with rocprofv3 --pmc in separate passes, NOT measured in this run)"
                               if traffic is not None else None),
            "avg_launch_ms": round(kdur * 1e3, 4), "alg_bytes_per_launch": alg_launch, "pass_ms": pass_ms,
            "transform_frac": round(6.0 * esz * E / world / dt_step / HBM_PEAK, 4),
            "transform_device_ms": round(dev_s * 1e3, 4)}
    return roof


def xgmi_block(group, esz, E, world, t_exchange_step):
    """per-link rate of an exchange inside a group of `group` ranks (None for a group of one: nothing leaves the GPU)"""
    if group <= 1 or t_exchange_step <= 0:
        return None
    link_bytes = esz * E / world / group  # per link and direction: (1/g) of the local volume to each of g-1 peers
    return {"group": group, "links_used": group - 1, "bytes_per_link_per_direction": link_bytes,
            "achieved_GBps_per_link_per_direction": round(link_bytes / t_exchange_step / 1e9, 2),
            "link_peak_GBps_bidirectional": XGMI_LINK / 1e9,
            "frac_of_link_bidirectional": round(2 * link_bytes / t_exchange_step / XGMI_LINK, 4)}


# ------------------------------------------------------------------------------------------------
# one rank
# ------------------------------------------------------------------------------------------------
def rank_main(args):
    import ctypes as C
    import faulthandler
    faulthandler.dump_traceback_later(args.launch_timeout, exit=True)  # a stuck rank ends with a traceback, not silently
    # stdout carries ONE line, the JSON record: the library prints its plan parameters there at init, as the reference does
    # (offt-compute.c:3416) -- they go to stderr for the length of this run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # HIP maps a process's streams onto 4 hardware queues by default and two streams on one queue run in order: a plan's
    # compute and comm streams must not end up sharing one (the exchange would no longer run under the passes).  Read by the
    # HIP runtime when it starts, hence before torch is imported; no effect on one GPU (16.94 against 16.96 ms)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    from offt_amd import api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    L = api.lib()
    prec = api.F64 if args.dtype == "f64" else api.F32
    esz = 16 if prec == api.F64 else 8
    tdt = torch.float64 if prec == api.F64 else torch.float32
    dist = None
    force_dist = bool(int(os.environ.get("OFFT_BENCH_FORCE_DIST", "0")))  # rehearse the multi-rank plumbing on one GPU

    def join_world(lib):
        """hand `lib` its world: rank 0 makes an RCCL id, torch.distributed ships the 128 bytes round"""
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            if lib.offt_hip_get_unique_id(buf):
                raise SystemExit("offt_hip_get_unique_id failed: " + lib.offt_hip_last_error().decode())
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        uid = uid.cuda()
        dist.broadcast(uid, src=0)
        idb = bytes(uid.cpu().numpy().tobytes())
        if lib.offt_hip_set_world(rank, world, idb, local_rank):
            raise SystemExit("offt_hip_set_world failed: " + lib.offt_hip_last_error().decode())

    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        join_world(L)
    multi = dist is not None

    n = args.n
    E = float(n) ** 3
    flops = 5.0 * E * math.log2(E)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_plan(p1):
        L = api.lib()
        params = {}
        if multi:
            params["P1"] = p1
        if args.layout == "xyz":
            params["S"] = 1
        po = api.offt_3d_init(n, n, n, custom_params=api.make_params(**params), precision=prec)
        data = torch.zeros(api.local_elems(po) * 2, dtype=tdt, device="cuda")
        torch.cuda.synchronize()  # the plan owns a non-blocking stream: order it after torch's zero fill
        L.offt_hip_fill_input(po, data.data_ptr(), 1)  # seeded position hash in [-1, 1)
        # keep magnitudes bounded over many back-to-back transforms: exact power-of-two rescale in the last store
        L.offt_hip_set_output_scale(po, 2.0 ** -(round(math.log2(E)) // 2))
        return po, data

    def timed(po, data, steps, warmup, skip=0, enqueue_only=False):
        """`steps` executes between two barriers: wall seconds (max over ranks), per-phase device seconds.
        enqueue_only (multi-rank headline): the K steps are enqueued back to back (offt_hip_set_async) and waited for
        once, as a caller that transforms a series of fields would do -- no host round trip between steps; the
        per-phase events exist only in the synchronous mode, so they come from a few extra steps afterwards."""
        ptr = data.data_ptr()
        L = api.lib()  # (the diagnostics build once the extras have switched to it)
        if skip:
            L.offt_hip_set_debug_skip(po, skip)
        for _ in range(warmup):
            api.offt_3d_execute(po, ptr, ptr)
        pass_acc, dev_acc = [0.0, 0.0, 0.0], 0.0
        if enqueue_only:
            L.offt_hip_set_async(po, 1)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                api.offt_3d_execute(po, ptr, ptr)
            if L.offt_hip_wait(po):
                raise RuntimeError("offt_hip_wait failed: " + L.offt_hip_last_error().decode())
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            L.offt_hip_set_async(po, 0)
            nph = max(2, min(4, steps))
            for _ in range(nph):
                api.offt_3d_execute(po, ptr, ptr)
                t3 = (C.c_double * 3)()
                L.offt_hip_last_pass_seconds(po, t3)
                for i in range(3):
                    pass_acc[i] += t3[i] * steps / nph
                dev_acc += L.offt_hip_last_device_seconds(po) * steps / nph
            barrier()
        else:
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                api.offt_3d_execute(po, ptr, ptr)  # returns after the GPU finished; per-pass HIP events inside
                t3 = (C.c_double * 3)()
                L.offt_hip_last_pass_seconds(po, t3)
                for i in range(3):
                    pass_acc[i] += t3[i]
                dev_acc += L.offt_hip_last_device_seconds(po)
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
        if skip:
            L.offt_hip_set_debug_skip(po, 0)
        return dt, [x / steps for x in pass_acc], dev_acc / steps

    # ---- headline: K timed steps ----
    p1_head = args.p1 if args.p1 > 0 else 1
    po, data = make_plan(p1_head)
    c = api.comm_dict(po)
    # the SAME mode at every N: one synchronous offt_3d_execute per step between two barriers (run-fft.c:374-395)
    dt, pass_s, dev_s = timed(po, data, args.steps, args.warmup, enqueue_only=False)
    steps = args.steps
    ms = dt / steps * 1e3
    value = flops * steps / dt / 1e9
    alg_bytes_transform = 6.0 * esz * E / world
    paired = int(L.offt_hip_last_passes_paired(po))

    def traffic_lookup(axis):
        # recorded for THIS workload only (1024^3 f64, z-y-x, one rank) and for the z pass, a single launch per transform:
        # the y / x records are per group launch (1/64 of a pass) and do not compare with alg_bytes_per_launch
        if not (os.path.exists(TRAFFIC_JSON) and world == 1 and n == 1024 and prec == api.F64 and not multi and args.layout == "zyx" and axis == "z"):
            return None
        try:
            return json.load(open(TRAFFIC_JSON)).get(axis + "_pass_hbm_bytes_per_launch")
        except Exception:
            return None
    roof = roofline_block(pass_s, paired, multi, esz, E, world, dt / steps, dev_s, traffic_lookup)
    cplx = "double-complex" if prec == api.F64 else "single-complex"
    out = {"metric": f"3D FFT GFLOP/s ({n}^3 {cplx} forward, 5*E*log2(E) flop model)",
           "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic (seeded position hash, device-resident)",
           "config": {"workload": f"{n}^3 {cplx} forward 3-D FFT, in-place, offt_3d_execute",
                      "grid": [n, n, n], "mesh": f"{c['p1']}x{c['p2']}", "output_layout": args.layout,
                      "steps_enqueued": "one synchronous call per step"},
           "roofline": roof}
    if multi:
        # second figure, NOT the headline: the K steps enqueued back to back with one wait, as a caller that transforms a
        # series of fields would do (no host round trip between steps)
        try:
            dt_p, _, _ = timed(po, data, steps, 1, enqueue_only=True)
            out["ms_per_step_sync"] = round(ms, 4)
            out["ms_per_step_pipelined"] = round(dt_p / steps * 1e3, 4)
            out["value_pipelined"] = round(flops * steps / dt_p / 1e9, 1)
        except Exception as e:
            out["ms_per_step_pipelined_error"] = repr(e)

    # ---- N > 1: what bounds the run?  (everything below is outside the headline's timed region) ----
    # The split timings need an execute that leaves out its passes or its exchanges: that switch exists only in the
    # DIAGNOSTICS build (tools/liboffthip_diag.so, `make`), never in the product library the headline was measured with.
    # The product's world is closed, the diagnostics build gets a world of its own, and the headline plan is made again there.
    diag_path = os.path.join(ROOT, "tools", "liboffthip_diag.so")
    if multi and not args.no_extras and not os.path.exists(diag_path):
        out["multi_gpu"] = {"n_ranks_seen": int(L.offt_hip_world_count()), "note": "tools/liboffthip_diag.so not built: no split timings"}
    elif multi and not args.no_extras:
        extra = {"library": "tools/liboffthip_diag.so (product + offt_hip_set_debug_skip); the headline above used offt_amd/liboffthip.so"}
        try:
            extra["n_ranks_seen"] = int(L.offt_hip_world_count())
            api.offt_3d_fin(po)
            po = None
            del data
            torch.cuda.empty_cache()
            barrier()
            L.offt_hip_finalize_world()
            L = api.use_library(diag_path)
            join_world(L)
            po, data = make_plan(p1_head)
            ksteps = max(2, min(5, steps))
            t_c, ps_c, _ = timed(po, data, ksteps, 1, skip=2)   # no exchanges
            t_x, _, _ = timed(po, data, ksteps, 1, skip=1)      # no FFT passes
            g = c["p2"] if c["p1"] == 1 else world
            extra["headline_split"] = {
                "compute_only_ms": round(t_c / ksteps * 1e3, 4), "exchange_only_ms": round(t_x / ksteps * 1e3, 4),
                "full_ms": round(ms, 4),
                "exposed_exchange_ms": round(max(0.0, ms - t_c / ksteps * 1e3), 4),
                "kernels_hbm_frac": round(alg_bytes_transform / (t_c / ksteps) / HBM_PEAK, 4),
                "xgmi": xgmi_block(g, esz, E, world, t_x / ksteps)}  # None for a group of one (one-GPU rehearsal)
        except Exception as e:  # extras never cost the headline
            extra["headline_split_error"] = repr(e)
        if po is not None:
            api.offt_3d_fin(po)
            po = None
            del data
        torch.cuda.empty_cache()
        try:
            # the reference's default mesh: the largest divisor of p that is <= sqrt(p) (2 x 4 at 8 ranks)
            p1_def = max(d for d in range(1, int(math.isqrt(world)) + 1) if world % d == 0)
            force_pencil = bool(int(os.environ.get("OFFT_BENCH_FORCE_PENCIL", "0")))  # one-GPU rehearsal of this branch
            if p1_def != p1_head or force_pencil:
                if force_pencil:
                    os.environ["OFFT_NO_SLAB_LAYOUT"] = "1"
                po2, data2 = make_plan(p1_def)
                c2 = api.comm_dict(po2)
                ksteps = max(2, min(5, steps))
                t_p, _, _ = timed(po2, data2, ksteps, 1)
                t_pc, _, _ = timed(po2, data2, ksteps, 1, skip=2)
                t_px, _, _ = timed(po2, data2, ksteps, 1, skip=1)
                extra["pencil_default_mesh"] = {
                    "mesh": f"{c2['p1']}x{c2['p2']}", "ms_per_step": round(t_p / ksteps * 1e3, 4),
                    "gflops": round(flops * ksteps / t_p / 1e9, 1), "compute_only_ms": round(t_pc / ksteps * 1e3, 4),
                    "exchange_only_ms": round(t_px / ksteps * 1e3, 4),
                    "bytes_per_link_per_direction": {"row_group": esz * E / world / c2["p2"], "column_group": esz * E / world / c2["p1"]},
                    "comm_streams": int(os.environ.get("OFFT_COMM_STREAMS", "1"))}
                api.offt_3d_fin(po2)
                del data2
                torch.cuda.empty_cache()
            else:
                extra["pencil_default_mesh"] = {"mesh": f"{p1_def}x{world // p1_def}", "note": "same as the headline mesh"}
        except Exception as e:
            extra["pencil_default_mesh_error"] = repr(e)
        try:
            probe = {}
            for label, mode, shift, nbytes in (("all_to_all_32MiB_per_peer", 0, 0, 32 << 20), ("ring_shift1_128MiB", 1, 1, 128 << 20)):
                barrier()
                sec = max_over_ranks(L.offt_hip_link_probe(mode, shift, nbytes, 3))
                probe[label] = {"seconds": round(sec, 6),  # (a world of one moves nothing: no rate)
                                "GBps_per_link_per_direction": round(nbytes / sec / 1e9, 2) if sec > 0 and world > 1 else None}
            extra["link_probe"] = probe
        except Exception as e:
            extra["link_probe_error"] = repr(e)
        out["multi_gpu"] = extra
    if po is not None:
        api.offt_3d_fin(po)

    if rank == 0:
        if world == 1 and not multi and not args.no_cpu_baseline:
            try:
                cores = len(os.sched_getaffinity(0))
                big = mem_available_gib() >= 64.0
                cn = args.cpu_n or (1024 if big else 512)
                out["cpu_baseline"] = cpu_baseline(cn, min(cores, 16), 1 if cn >= 1024 else 2)
                out["cpu_baseline"]["sample"] += f" (host MemAvailable {mem_available_gib():.0f} GiB)"
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        api.lib().offt_hip_finalize_world()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args, argv))  # parent: spawn, watch, relay -- before anything touches the GPU
    if args.launch_selftest:
        if "WORLD_SIZE" not in os.environ:  # --gpus 1: a world of one
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        return selftest_rank(args)
    rank_main(args)


if __name__ == "__main__":
    main()
