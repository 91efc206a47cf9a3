"""One rank of the shared-GPU multi-rank test (launched by test_gpu_world.py): real HIP kernels and device
buffers, the exchange host-staged over gloo because RCCL will not put two ranks on one device."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    import torch
    import torch.distributed as dist
    rank, size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    cfg = json.loads(sys.argv[1])
    outdir = sys.argv[2]
    dist.init_process_group("gloo", rank=rank, world_size=size)
    torch.cuda.set_device(0)
    import oracle_lib as O
    import cpu_world
    from cpu_world import A2A_CB
    from offt_amd import api
    L = cpu_world.test_lib()  # the build with the test-only transport seam; same kernels and host code as the product
    state = {"p1": 1}

    def transport(which, npeers, peer_in_group, sendp, sendbytes, recvp, recvbytes):
        try:
            reqs, pend = [], []
            for a in range(npeers):
                peer = cpu_world.group_peer(which, peer_in_group[a], rank, size, L.offt_hip_test_current_p1())
                sb, rb = sendbytes[a], recvbytes[a]
                if peer == rank:
                    if sb:
                        assert sb == rb
                        tmp = (C.c_char * sb)()
                        L.offt_hip_memcpy_d2h(tmp, sendp[a], sb)
                        L.offt_hip_memcpy_h2d(recvp[a], tmp, sb)
                    continue
                if rb:
                    t = torch.empty(rb, dtype=torch.uint8)
                    pend.append((t, recvp[a], rb))
                    reqs.append(dist.irecv(t, src=peer, tag=which))
                if sb:
                    t = torch.empty(sb, dtype=torch.uint8)
                    L.offt_hip_memcpy_d2h(C.c_void_p(t.data_ptr()), sendp[a], sb)
                    pend.append((t, None, 0))
                    reqs.append(dist.isend(t, dst=peer, tag=which))
            for r in reqs:
                r.wait()
            for t, dst, nb in pend:
                if dst is not None:
                    L.offt_hip_memcpy_h2d(dst, C.c_void_p(t.data_ptr()), nb)
            return 0
        except Exception as e:
            print("transport failed:", repr(e), flush=True)
            return -1

    cb = A2A_CB(transport)
    for ci, case in enumerate(cfg):
        shape = case["N"]
        p1 = case["params"].get("P1") or O.params_default(*shape, size)[0]
        state["p1"] = p1
        L.offt_hip_test_set_transport(C.cast(cb, C.c_void_p), rank, size)
        r2c = case.get("r2c", 0)
        if case.get("p2p"):
            # direct-store exchange between PROCESSES: the product's own hipIpc path (handles gathered through the
            # transport above, opened with hipIpcOpenMemHandle) -- the kernels of one process store into another's buffers
            os.environ["OFFT_EXCHANGE"] = "p2p"
        po = api.offt_3d_init(*shape, custom_params=api.make_params(**case["params"]), is_equalxy=case.get("eq", 0), is_r2c=r2c)
        os.environ.pop("OFFT_EXCHANGE", None)
        L.offt_hip_get_exchange.argtypes = [C.c_void_p]
        exchange = L.offt_hip_get_exchange(po)
        c = api.comm_dict(po)
        v = list(po.contents.params.contents.v)
        dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        if L.offt_hip_fill_input(po, dev.data_ptr(), 1):
            raise SystemExit("fill failed")
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        buf = dev.cpu().numpy().view(np.complex128)
        np.save(os.path.join(outdir, f"case{ci}_rank{rank}.npy"), buf)
        for _ in range(case.get("repeat", 0)):  # the same plan again: buffer reuse between transforms
            if L.offt_hip_fill_input(po, dev.data_ptr(), 1):
                raise SystemExit("fill failed")
            api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
            if not np.array_equal(dev.cpu().numpy().view(np.complex128), buf):
                raise SystemExit("repeat: result differs from the first transform")
        if case.get("inv"):
            torch.cuda.synchronize()
            api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
            import cpu_world
            np.save(os.path.join(outdir, f"case{ci}_rank{rank}_inv.npy"),
                    cpu_world.input_block(c, dev.cpu().numpy().view(np.complex128)))
        api.offt_3d_fin(po)
        json.dump({"comm": c, "v": v, "exchange": exchange}, open(os.path.join(outdir, f"case{ci}_rank{rank}.json"), "w"))
        dist.barrier()
    L.offt_hip_test_set_transport(None, 0, 1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
