"""Several OFFT ranks as THREADS of one process sharing the one GPU of the test box (launched by
test_gpu_world.py).  The box admits at most 6 processes on the card, so the 8-rank meshes the multi-GPU bench uses
(1x8, 2x4, 8x1) cannot be rehearsed as 8 processes; the test build's seam state is per thread instead.  Every kernel
launch, descriptor, device buffer, stream and event is the product's; only the transport is swapped: a rank "sends"
by posting its device pointer, the receiver copies device-to-device (RCCL refuses two ranks on one device).

usage: _gpu_thread_world.py <size> <cases.json> <outdir>
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


def main():
    size, cases, outdir = int(sys.argv[1]), json.loads(sys.argv[2]), sys.argv[3]
    import torch
    torch.cuda.set_device(0)
    import cpu_world
    import oracle_lib as O
    from offt_amd import api
    L = cpu_world.test_lib()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    wire = Wire()
    results, errors = {}, []

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
                        if hip.hipMemcpy(recvp[a], ptr, nb, 3) != 0:  # hipMemcpyDeviceToDevice
                            raise RuntimeError("hipMemcpy failed")
                        L.offt_hip_device_synchronize()  # the plan's streams are non-blocking: finish the copy before releasing
                        ack.set()
                for ack in acks:
                    if not ack.wait(120.0):
                        raise RuntimeError("send not consumed")
                return 0
            except Exception as e:
                print("transport failed on rank", rank, repr(e), flush=True)
                wire.fail()
                return -1
        return cpu_world.A2A_CB(transport)

    def rank_thread(rank, ci, case, bar):
        try:
            cb = make_transport(rank)
            L.offt_hip_test_set_transport(C.cast(cb, C.c_void_p), rank, size)
            prec = api.F32 if case.get("f32") else api.F64
            r2c = case.get("r2c", 0)
            po = api.offt_3d_init(*case["N"], custom_params=api.make_params(**case["params"]), is_equalxy=case.get("eq", 0),
                                  is_r2c=r2c, precision=prec)
            c = api.comm_dict(po)
            v = list(po.contents.params.contents.v)
            dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
            torch.cuda.synchronize()
            ramp = case.get("check") == "ramp"
            if L.offt_hip_fill_input(po, dev.data_ptr(), 0 if ramp else 1):
                raise RuntimeError("fill failed")
            ct = np.complex128 if prec == api.F64 else np.complex64
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
                L.offt_hip_test_set_transport(None, 0, 1)
                results[(ci, rank)] = {"comm": c, "v": v, "e_in": e_in, "e_out": e_out, "spots": spots}
                return
            bar.wait()
            api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            res = {"comm": c, "v": v, "out": dev.cpu().numpy().view(ct).copy()}
            if case.get("inv"):
                bar.wait()
                api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
                res["inv"] = cpu_world.input_block(c, dev.cpu().numpy().view(ct))
            bar.wait()
            api.offt_3d_fin(po)
            L.offt_hip_test_set_transport(None, 0, 1)
            results[(ci, rank)] = res
        except Exception as e:
            errors.append((ci, rank, repr(e)))
            wire.fail()
            try:
                bar.abort()
            except Exception:
                pass

    summary = []
    for ci, case in enumerate(cases):
        for k, val in case.get("env", {}).items():
            os.environ[k] = str(val)
        bar = threading.Barrier(size)
        th = [threading.Thread(target=rank_thread, args=(r, ci, case, bar)) for r in range(size)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for k in case.get("env", {}):
            os.environ.pop(k, None)
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
