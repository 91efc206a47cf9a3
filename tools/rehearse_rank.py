#!/usr/bin/env python3
"""One rank's share of a multi-GPU transform, rehearsed on ONE GPU: the plan is built as rank 0 of `--ranks` ranks (true
decomposition, true per-rank buffers, true pass descriptors), the exchanges are no-ops (a test transport that moves
nothing), so the kernels K1 / K2 / K3 run on exactly the shapes and strides they have on an 8-GPU node.  Results are
garbage, durations are real: compute-only time per rank and -- under rocprofv3 --kernel-trace -- per-kernel averages.

  python tools/rehearse_rank.py --n 1024 --ranks 8 --p1 1 [--dtype f64] [--reps 5]
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rehearse -- python3 tools/rehearse_rank.py ...
With OFFT_LOG_PASSES=1 in the environment the library prints one `offt-pass` line per launch (elements per launch);
tools/summarize_rehearsal.py joins them with the kernel trace."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--p1", type=int, default=1)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--layout", default="zyx")
    ap.add_argument("--t1", type=int, default=-1, help="x-tile thickness (default: the library's)")
    ap.add_argument("--t2", type=int, default=-1, help="z-chunk thickness (default: the library's)")
    ap.add_argument("--inverse", type=int, default=0, help="1: time offt_3d_execute_dir(+1) (the mirrored schedule) instead of the forward transform")
    ap.add_argument("--touch", type=int, default=0, help="1: the first transform really fills the receive volumes (see the transport below)")
    ap.add_argument("--variants", default="", help="kernel variants vx,vy,vz (offt_hip_set_variant; 200 + id = column-pair variant id)")
    args = ap.parse_args()
    os.environ.setdefault("OFFT_TEST_TRANSPORT_NOSYNC", "1")  # the no-op exchange needs no host synchronisation
    import torch
    torch.cuda.set_device(0)
    import cpu_world
    from offt_amd import api
    L = cpu_world.test_lib()
    # the exchange moves nothing -- except, with --touch, during the FIRST transform, when every receive block is really
    # written (device-to-device from the matching send block).  Without that the receive volume stays memory nobody has
    # ever stored to, and reads of such memory are served faster than reads of real data: K2, and through the Infinity
    # Cache K3, then look 10-15 % better than they are (profiles/r03_rehearse_touch.txt)
    state = {"calls": 0, "touch": args.touch}
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def transport(which, npeers, peer, sendp, sendbytes, recvp, recvbytes):
        if state["touch"]:
            for a in range(npeers):
                if recvbytes[a] and sendbytes[a] >= recvbytes[a]:
                    hip.hipMemcpy(recvp[a], sendp[a], recvbytes[a], 3)
        return 0
    cb = cpu_world.A2A_CB(transport)
    L.offt_hip_test_set_transport(C.cast(cb, C.c_void_p), 0, args.ranks)
    prec = api.F64 if args.dtype == "f64" else api.F32
    esz = 16 if prec == api.F64 else 8
    params = dict(P1=args.p1)
    if args.layout == "xyz":
        params["S"] = 1
    if args.t1 > 0:
        params["T1"] = args.t1
    if args.t2 > 0:
        params["T2"] = args.t2
    n = args.n
    po = api.offt_3d_init(n, n, n, custom_params=api.make_params(**params), precision=prec)
    c = api.comm_dict(po)
    if args.variants:
        for ax, vv in enumerate(int(x) for x in args.variants.split(",")):
            L.offt_hip_set_variant(po, ax, vv)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64 if prec == api.F64 else torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    L.offt_hip_set_output_scale(po, 2.0 ** -16)
    best = None
    if args.touch:
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())  # every receive block written once
        state["touch"] = False
    for _ in range(args.reps):
        if args.inverse:
            api.offt_3d_execute_dir(po, dev.data_ptr(), dev.data_ptr(), +1)
        else:
            api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
        t = (C.c_double * 3)()
        L.offt_hip_last_pass_seconds(po, t)
        tot = L.offt_hip_last_device_seconds(po)
        if best is None or tot < best[0]:
            best = (tot, list(t))
    local = float(n) ** 3 / args.ranks
    v = list(po.contents.params.contents.v)
    print(f"rehearsal {'INVERSE ' if args.inverse else ''}{n}^3 {args.dtype} rank 0 of {args.ranks}, mesh {c['p1']}x{c['p2']}, variants {args.variants or 'default'}, T1 {v[1]} W1 {v[2]} T2 {v[12]}: kernels only "
          f"{best[0]*1e3:.3f} ms per transform = {6*esz*local/best[0]/8e12*100:.1f} % of 8 TB/s on the rank's 6*S*E/P bytes "
          f"(phase 1 {best[1][0]*1e3:.3f} ms, last phase {best[1][2]*1e3:.3f} ms)", flush=True)
    api.offt_3d_fin(po)
    L.offt_hip_test_set_transport(None, 0, 1)


if __name__ == "__main__":
    main()
