"""Several OFFT ranks as THREADS of one process (launched by test_gpu_world.py and test_p2p_world.py).

backend gpu: the ranks share the one GPU of the test box.  The box admits at most 6 processes on the card, so the 8-rank
meshes the multi-GPU bench uses (1x8, 2x4, 8x1) cannot be rehearsed as 8 processes; the test build's seam state is per
thread instead.  Every kernel launch, descriptor, device buffer, stream and event is the product's; only the transport is
swapped: a rank "sends" by posting its device pointer, the receiver copies device-to-device (RCCL refuses two ranks on
one device).
backend cpu: the same host logic on the test-only CPU descriptor interpreter (tests/cpu_backend.c), host memory.

A case with "async": 1 (gpu) runs the staged exchange through an ASYNCHRONOUS transport: nothing is drained on the host --
the sender records an event on its comm stream, the receiver makes ITS comm stream wait for that event, enqueues the
device-to-device copy there and records an event of its own, which the sender's comm stream waits for before it goes on
(a send "completes" when the data has left the buffer, as with RCCL).  The host only hands pointers and events round;
on the device the schedules' own event edges between compute and comm streams are all that orders kernels and copies, so a
missing edge shows up as a wrong result instead of being hidden by a host synchronisation.

A case with "p2p": 1 runs the DIRECT-STORE exchange (OFFT_EXCHANGE=p2p): no transport at all for the data -- the packing
kernels of one rank store straight into the other ranks' receive volumes (another thread's buffer: a peer's memory) and
flag kernels order the streams.  hipIpc cannot open a handle inside the process that made it, so peer_open goes through a
test seam that hands the threads' pointers round; the hook seam is a barrier among the rank threads before every wait is
enqueued (offt_backend.h).

usage: _thread_world.py <size> <cases.json> <outdir> [gpu|cpu]
"""
import collections
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


class Wire:
    """FIFO per (which, src, dst): the sender posts (pointer, bytes), the receiver copies and acknowledges"""

    def __init__(self):
        self.cv = threading.Condition()
        self.q = collections.defaultdict(collections.deque)
        self.failed = False

    def post(self, key, ptr, nbytes):
        ack = threading.Event()
        with self.cv:
            self.q[key].append((ptr, nbytes, ack))
            self.cv.notify_all()
        return ack

    def take(self, key, timeout=120.0):
        with self.cv:
            ok = self.cv.wait_for(lambda: self.q[key] or self.failed, timeout)
            if not ok or self.failed:
                raise RuntimeError(f"no message on {key}")
            return self.q[key].popleft()

    def fail(self):
        with self.cv:
            self.failed = True
            self.cv.notify_all()


PEER_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p))
HOOK_CB = C.CFUNCTYPE(None)


def main():
    size, cases, outdir = int(sys.argv[1]), json.loads(sys.argv[2]), sys.argv[3]
    cpu = len(sys.argv) > 4 and sys.argv[4] == "cpu"
    import cpu_world
    import oracle_lib as O
    from offt_amd import api
    L = cpu_world.test_lib()
    L.offt_hip_test_set_p2p.argtypes = [C.c_void_p, C.c_void_p]
    L.offt_hip_test_set_p2p.restype = None
    L.offt_hip_get_exchange.argtypes = [C.c_void_p]
    if cpu:
        torch = None
        CB = cpu_world._cb_lib()
        CB.cpu_backend_set_peer_open.argtypes = [C.c_void_p]
    else:
        import torch
        torch.cuda.set_device(0)
        hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        hip.hipEventDestroy.argtypes = [C.c_void_p]
        L.offt_hip_test_set_transport_async.argtypes = [C.c_void_p]
        L.offt_hip_test_set_transport_async.restype = None
    ASYNC_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                           C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)
    wire = Wire()
    results, errors = {}, []
    tls = threading.local()   # the CPU backend's callbacks are process-wide: the calling thread knows its rank
    hook_bar = {"b": None}

    def copy(dst, src, nb):
        if cpu:
            C.memmove(dst, src, nb)
        else:
            if hip.hipMemcpy(dst, src, nb, 3) != 0:  # hipMemcpyDeviceToDevice
                raise RuntimeError("hipMemcpy failed")
            L.offt_hip_device_synchronize()  # the plan's streams are non-blocking: finish the copy before releasing

    # ---- direct-store exchange seams ----
    peer_book = {"cv": threading.Condition(), "posted": {}, "count": collections.defaultdict(int)}

    def peer_open(which, npeers, self_idx, local, nbytes, peers):
        """rendezvous of one group's members: everybody posts its pointer, everybody reads the others'"""
        try:
            rank = tls.rank
            p1 = L.offt_hip_test_current_p1()
            members = [cpu_world.group_peer(which, g, rank, size, p1) for g in range(npeers)]
            assert members[self_idx] == rank
            gid = (which, members[0])
            with peer_book["cv"]:
                seq = peer_book["count"][(gid, rank)]
                peer_book["count"][(gid, rank)] += 1
                peer_book["posted"][(gid, seq, rank)] = (local, nbytes)
                peer_book["cv"].notify_all()
                ok = peer_book["cv"].wait_for(lambda: all((gid, seq, m) in peer_book["posted"] for m in members) or wire.failed, 120.0)
                if not ok or wire.failed:
                    return -1
                for g, m in enumerate(members):
                    ptr, nb = peer_book["posted"][(gid, seq, m)]
                    assert nb == nbytes
                    peers[g] = ptr
            return 0
        except Exception as e:
            print("peer_open failed:", repr(e), flush=True)
            wire.fail()
            return -1

    def hook():
        try:
            hook_bar["b"].wait(120.0)
        except threading.BrokenBarrierError:
            wire.fail()

    peer_cb, hook_cb = PEER_CB(peer_open), HOOK_CB(hook)

    def make_transport(rank):
        def transport(which, npeers, peer_in_group, sendp, sendbytes, recvp, recvbytes):
            try:
                p1 = L.offt_hip_test_current_p1()
                acks = []
                for a in range(npeers):
                    if sendbytes[a]:
                        peer = cpu_world.group_peer(which, peer_in_group[a], rank, size, p1)
                        acks.append(wire.post((which, rank, peer), sendp[a], sendbytes[a]))
                for a in range(npeers):
                    if recvbytes[a]:
                        peer = cpu_world.group_peer(which, peer_in_group[a], rank, size, p1)
                        ptr, nb, ack = wire.take((which, peer, rank))
                        assert nb == recvbytes[a], (which, peer, rank, nb, recvbytes[a])
                        copy(recvp[a], ptr, nb)
                        ack.set()
                for ack in acks:
                    if not ack.wait(120.0):
                        raise RuntimeError("send not consumed")
                return 0
            except Exception as e:
                print("transport failed on rank", rank, repr(e), flush=True)
                wire.fail()
                return -1
        return transport

    garbage = []  # events stay alive until the case is over (another rank's stream may still wait on them)

    def new_event(stream):
        ev = C.c_void_p()
        if hip.hipEventCreateWithFlags(C.byref(ev), 2) != 0 or hip.hipEventRecord(ev, stream) != 0:  # hipEventDisableTiming
            raise RuntimeError("event")
        garbage.append(ev)
        return ev

    def make_async_transport(rank):
        def transport(which, npeers, peer_in_group, sendp, sendbytes, recvp, recvbytes, stream):
            try:
                p1 = L.offt_hip_test_current_p1()
                ready = new_event(stream)  # everything this rank's comm stream waited for (the packing kernel) lies before it
                # a SLOW wire: a millisecond of spinning on this rank's comm stream ahead of its copies.  The host runs ahead
                # of the device, so a kernel that does not wait for its exchange really starts before the data is there
                # (tools/async_negative_control.sh: with the edge "K2 waits for its chunk" dropped the result is wrong)
                # ("async": 2 leaves the wire fast: together with passes that are made slow -- OFFT_TEST_SLOW_PASS_MS, test
                # build -- an exchange that does not wait for the kernel packing its data reads too early)
                if tls.async_mode == 1:
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                        torch.cuda._sleep(2_000_000)
                acks = []
                for a in range(npeers):
                    if sendbytes[a]:
                        peer = cpu_world.group_peer(which, peer_in_group[a], rank, size, p1)
                        box = {}
                        acks.append((wire.post((which, rank, peer), (sendp[a], ready, box), sendbytes[a]), box))
                for a in range(npeers):
                    if recvbytes[a]:
                        peer = cpu_world.group_peer(which, peer_in_group[a], rank, size, p1)
                        (ptr, peer_ready, box), nb, ack = wire.take((which, peer, rank))
                        assert nb == recvbytes[a], (which, peer, rank, nb, recvbytes[a])
                        if os.environ.get("OFFT_TEST_ASYNC_NOCOPY"):  # (sensitivity check of the harness itself: nothing arrives)
                            pass
                        elif hip.hipStreamWaitEvent(stream, peer_ready, 0) != 0 or hip.hipMemcpyAsync(recvp[a], ptr, nb, 3, stream) != 0:
                            raise RuntimeError("async copy")
                        box["done"] = new_event(stream)  # the sender's buffer has been read once this event has passed
                        ack.set()
                for ack, box in acks:
                    if not ack.wait(120.0):
                        raise RuntimeError("send not consumed")
                    if hip.hipStreamWaitEvent(stream, box["done"], 0) != 0:  # my send is complete when the receiver has copied
                        raise RuntimeError("wait")
                return 0
            except Exception as e:
                print("async transport failed on rank", rank, repr(e), flush=True)
                wire.fail()
                return -1
        return transport

    transports = {r: make_transport(r) for r in range(size)}
    async_cbs = {}
    cpu_a2a = cpu_world.A2A_CB(lambda *a: transports[tls.rank](*a))  # CPU backend: ONE process-wide callback

    def unseam():
        L.offt_hip_test_set_p2p(None, None)
        if cpu:
            L.offt_hip_test_set_backend(None, 0, 1)
        else:
            L.offt_hip_test_set_transport_async(None)
            L.offt_hip_test_set_transport(None, 0, 1)

    def rank_thread(rank, ci, case, bar):
        try:
            tls.rank = rank
            if cpu:
                L.offt_hip_test_set_backend(CB.cpu_backend_table(), rank, size)
            else:
                cb = cpu_world.A2A_CB(transports[rank])
                L.offt_hip_test_set_transport(C.cast(cb, C.c_void_p), rank, size)
                if case.get("async"):
                    tls.async_mode = int(case["async"])
                    async_cbs[rank] = ASYNC_CB(make_async_transport(rank))
                    L.offt_hip_test_set_transport_async(C.cast(async_cbs[rank], C.c_void_p))
            if case.get("p2p"):
                L.offt_hip_test_set_p2p(C.cast(peer_cb, C.c_void_p), C.cast(hook_cb, C.c_void_p))
            prec = api.F32 if case.get("f32") else api.F64
            r2c = case.get("r2c", 0)
            po = api.offt_3d_init(*case["N"], custom_params=api.make_params(**case["params"]), is_equalxy=case.get("eq", 0),
                                  is_r2c=r2c, precision=prec)
            c = api.comm_dict(po)
            v = list(po.contents.params.contents.v)
            if case.get("p2p") and L.offt_hip_get_exchange(po) != 1:
                raise RuntimeError("the plan fell back to the staged exchange")
            ct = np.complex128 if prec == api.F64 else np.complex64
            if cpu:
                return cpu_rank(rank, ci, case, bar, po, c, v, ct)
            dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
            torch.cuda.synchronize()
            ramp = case.get("check") == "ramp"
            if L.offt_hip_fill_input(po, dev.data_ptr(), 0 if ramp else 1):
                raise RuntimeError("fill failed")
            if ramp:
                # full-size property check (no host copy of the grid): energy before / after, closed-form spot values
                torch.cuda.synchronize()
                e_in = float(dev.double().square().sum())
                bar.wait()
                api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
                e_out = float(dev.double().square().sum())
                cv = torch.view_as_complex(dev.view(-1, 2))
                spots = {}
                os_, oz, ost = c["ostart"], c["osize"], c["ostride"]
                for g in case["spots"]:
                    loc = [g[d] - os_[d] for d in range(3)]
                    if all(0 <= loc[d] < oz[d] for d in range(3)):
                        z = complex(cv[loc[0] * ost[0] + loc[1] * ost[1] + loc[2] * ost[2]])
                        spots[",".join(map(str, g))] = [z.real, z.imag]
                bar.wait()
                api.offt_3d_fin(po)
                unseam()
                results[(ci, rank)] = {"comm": c, "v": v, "e_in": e_in, "e_out": e_out, "spots": spots}
                return
            bar.wait()
            api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            res = {"comm": c, "v": v, "out": dev.cpu().numpy().view(ct).copy()}
            for rep_i in range(case.get("repeat", 0)):
                # the same plan again on OTHER input (buffer reuse between transforms: the FREE flags of the direct-store
                # exchange; a block of the previous transform read in place of this one's would show): the field times
                # 2^(i+1), exactly representable, so the result is the first one times 2^(i+1), bit for bit
                if L.offt_hip_fill_input(po, dev.data_ptr(), 1):
                    raise RuntimeError("fill failed")
                dev.mul_(2.0 ** (rep_i + 1))
                torch.cuda.current_stream().synchronize()  # (this thread's stream only: a device-wide wait would line the ranks up)
                api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
                again = dev.cpu().numpy().view(ct)
                if not np.array_equal(again, res["out"] * ct(2.0 ** (rep_i + 1))):
                    raise RuntimeError(f"repeat {rep_i}: result differs from the first transform (times {2 ** (rep_i + 1)})")
            if case.get("inv"):
                bar.wait()
                api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
                # (dev holds the LAST forward result: the field times 2^repeat)
                res["inv"] = cpu_world.input_block(c, dev.cpu().numpy().view(ct)) / ct(2.0 ** case.get("repeat", 0))
                if case.get("repeat"):
                    if L.offt_hip_fill_input(po, dev.data_ptr(), 1):
                        raise RuntimeError("fill failed")
                    dev.mul_(0.5)
                    torch.cuda.current_stream().synchronize()
                    api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
                    if not np.array_equal(dev.cpu().numpy().view(ct), res["out"] * ct(0.5)):
                        raise RuntimeError("forward after inverse differs from the first transform (times 1/2)")
                    if case.get("inv") == 2:  # the inverse again, on the same array: the schedule kept from the first one, other data
                        api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
                        if not np.array_equal(cpu_world.input_block(c, dev.cpu().numpy().view(ct)), res["inv"] * ct(0.5)):
                            raise RuntimeError("second inverse (kept schedule) differs from the first (times 1/2)")
            bar.wait()
            api.offt_3d_fin(po)
            unseam()
            results[(ci, rank)] = res
        except Exception as e:
            errors.append((ci, rank, repr(e)))
            wire.fail()
            for b in (bar, hook_bar["b"]):
                try:
                    b.abort()
                except Exception:
                    pass

    def cpu_rank(rank, ci, case, bar, po, c, v, ct):
        """host arrays on the CPU descriptor interpreter: fill through istart/isize/istride, execute, (inverse), repeat"""
        shape = case["N"]

        def fill():
            buf = np.zeros(api.local_elems(po), dtype=ct)
            i0, i1, i2 = c["isize"]
            if i0 and i1 and i2:
                f = O.hash_field(i0, i1, i2, *c["istart"])
                s0, s1, s2 = c["istride"]
                if case.get("r2c"):  # real rows: scalar index z + 2*istride1*y + 2*istride0*x (run-fft.c:54)
                    rv = buf.view(np.float64 if ct == np.complex128 else np.float32)
                    idx = np.arange(i0)[:, None, None] * 2 * s0 + np.arange(i1)[None, :, None] * 2 * s1 + np.arange(i2)[None, None, :]
                    rv[idx.ravel()] = f.real.ravel()
                else:
                    idx = np.arange(i0)[:, None, None] * s0 + np.arange(i1)[None, :, None] * s1 + np.arange(i2)[None, None, :] * s2
                    buf[idx.ravel()] = f.astype(ct).ravel()
            return buf
        buf = fill()
        ptr = buf.ctypes.data_as(C.c_void_p)
        bar.wait()
        api.offt_3d_execute(po, ptr, ptr)
        res = {"comm": c, "v": v, "out": buf.copy()}
        for rep_i in range(case.get("repeat", 0)):
            b2 = fill() * ct(2.0 ** (rep_i + 1))  # other data every time: see the GPU path
            p2 = b2.ctypes.data_as(C.c_void_p)
            api.offt_3d_execute(po, p2, p2)
            if not np.array_equal(b2, res["out"] * ct(2.0 ** (rep_i + 1))):
                raise RuntimeError(f"repeat {rep_i}: result differs from the first transform (times {2 ** (rep_i + 1)})")
        if case.get("inv"):
            back = buf.copy()
            bp = back.ctypes.data_as(C.c_void_p)
            api.offt_3d_execute_dir(po, bp, bp, +1)
            res["inv"] = cpu_world.input_block(c, back)
            if case.get("inv") == 2:  # once more on the SAME array: the plan replays the schedule it kept from the first inverse
                back[:] = buf
                api.offt_3d_execute_dir(po, bp, bp, +1)
                if not np.array_equal(cpu_world.input_block(c, back), res["inv"]):
                    raise RuntimeError("second inverse (kept schedule) differs from the first")
            if case.get("repeat"):
                b2 = fill()
                p2 = b2.ctypes.data_as(C.c_void_p)
                api.offt_3d_execute(po, p2, p2)
                if not np.array_equal(b2, res["out"]):
                    raise RuntimeError("forward after inverse differs from the first transform")
        bar.wait()
        api.offt_3d_fin(po)
        unseam()
        results[(ci, rank)] = res

    if cpu:
        CB.cpu_backend_set_a2a(cpu_a2a)
        CB.cpu_backend_set_peer_open(C.cast(peer_cb, C.c_void_p))
    summary = []
    for ci, case in enumerate(cases):
        for k, val in case.get("env", {}).items():
            os.environ[k] = str(val)
        if case.get("p2p"):
            os.environ["OFFT_EXCHANGE"] = "p2p"
        bar = threading.Barrier(size)
        hook_bar["b"] = threading.Barrier(size)
        peer_book["posted"].clear()
        peer_book["count"].clear()
        th = [threading.Thread(target=rank_thread, args=(r, ci, case, bar)) for r in range(size)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in case.get("env", {}):
            os.environ.pop(k, None)
        os.environ.pop("OFFT_EXCHANGE", None)
        if not cpu:
            torch.cuda.synchronize()
            for ev in garbage:
                hip.hipEventDestroy(ev)
            del garbage[:]
        if errors:
            print("FAILED", errors, flush=True)
            sys.exit(1)
        shape = tuple(case["N"])
        r2c = case.get("r2c", 0)
        f32 = bool(case.get("f32"))
        if case.get("check") == "ramp":
            E = float(np.prod(shape))
            e_in = sum(results[(ci, r)]["e_in"] for r in range(size))
            e_out = sum(results[(ci, r)]["e_out"] for r in range(size))
            n = shape[0]
            worst = abs(e_out / (E * e_in) - 1.0)   # Parseval, unnormalised forward transform
            got = {}
            for r in range(size):
                got.update(results[(ci, r)]["spots"])
            assert len(got) == len(case["spots"]), (sorted(got), case["spots"])
            for key, (re, im) in got.items():
                gx, gy, gz = map(int, key.split(","))
                nz_axes = [(gx, 100.0), (gy, 10.0), (gz, 1.0)]
                k = [(kk, w) for kk, w in nz_axes if kk]
                if not k:
                    want = n ** 3 * 111 * (n - 1) / 2
                elif len(k) == 1:
                    want = k[0][1] * n ** 3 * (-0.5 + 0.5j / np.tan(np.pi * k[0][0] / n))
                else:
                    want = 0.0
                scale = abs(want) if want else n ** 3 * 111 * (n - 1) / 2
                worst = max(worst, abs(complex(re, im) - want) / scale)
            rec = {"case": case, "mesh": [results[(ci, 0)]["comm"]["p1"], results[(ci, 0)]["comm"]["p2"]], "rel_numpy": worst,
                   "v": results[(ci, 0)]["v"], "tol": 5e-6 if f32 else 1e-12}
            summary.append(rec)
            print(json.dumps({k: rec[k] for k in rec if k != "v"}), flush=True)
            continue
        oshape = (shape[0], shape[1], shape[2] // 2 + 1) if r2c else shape
        G = np.full(oshape, np.nan + 0j)
        for r in range(size):
            cpu_world.scatter_out(results[(ci, r)]["comm"], results[(ci, r)]["out"], G)
        assert not np.isnan(G).any(), case
        field = O.hash_field(*shape)
        want = np.fft.rfftn(field.real, axes=(0, 1, 2)) if r2c else np.fft.fftn(field)
        e_np = float(np.linalg.norm(G - want) / np.linalg.norm(want))
        rec = {"case": case, "mesh": [results[(ci, 0)]["comm"]["p1"], results[(ci, 0)]["comm"]["p2"]], "rel_numpy": e_np,
               "v": results[(ci, 0)]["v"]}
        if np.prod(shape) <= 1 << 22:  # the oracle finishes these in seconds
            og, _, _ = O.world_fft(*shape, size, kind=1, is_equalxy=case.get("eq", 0), is_r2c=r2c, **case["params"])
            rec["rel_oracle"] = float(np.linalg.norm(G - og) / np.linalg.norm(og))
        if case.get("inv"):
            worst = 0.0
            for r in range(size):
                m = results[(ci, r)]["comm"]
                blk = O.hash_field(*m["isize"], *m["istart"])
                if blk.size:
                    worst = max(worst, float(np.linalg.norm(results[(ci, r)]["inv"] / np.prod(shape) - blk) / np.linalg.norm(blk)))
            rec["rel_inverse"] = worst
        rec["tol"] = 5e-6 if f32 else 1e-13
        summary.append(rec)
        print(json.dumps({k: rec[k] for k in rec if k != "v"}), flush=True)
    json.dump(summary, open(os.path.join(outdir, "summary.json"), "w"))


if __name__ == "__main__":
    main()
