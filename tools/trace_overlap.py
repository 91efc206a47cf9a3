#!/usr/bin/env python3
"""Timeline of ONE transform of a multi-rank schedule with REAL RCCL exchanges on one GPU: the plan runs as rank 0 of a
one-rank RCCL world with the tile pipeline and the exchanges forced on (every tile goes through ncclSend/ncclRecv to
self on the comm stream).  Under rocprofv3 --kernel-trace the summary shows the FFT launches of the compute stream
interleaved with the RCCL kernels of the comm stream -- the overlap the schedules are built for.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/overlap -- python3 tools/trace_overlap.py --schedule pencil
  python3 tools/trace_overlap.py --summarize gpurun_out/overlap > profiles/r02_overlap_trace_pencil.txt"""
import argparse
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    os.environ["OFFT_FORCE_PIPELINE"] = "1"
    os.environ["OFFT_FORCE_A2A"] = "1"
    if args.schedule == "pencil":
        os.environ["OFFT_NO_SLAB_LAYOUT"] = "1"
    if args.comm_streams:
        os.environ["OFFT_COMM_STREAMS"] = str(args.comm_streams)
    import torch
    from offt_amd import api
    torch.cuda.set_device(0)
    L = api.lib()
    uid = (C.c_char * 128)()
    assert L.offt_hip_get_unique_id(uid) == 0
    assert L.offt_hip_set_world(0, 1, uid, 0) == 0
    n = args.n
    po = api.offt_3d_init(n, n, n)
    dev = torch.zeros(api.local_elems(po) * 2, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    L.offt_hip_fill_input(po, dev.data_ptr(), 1)
    L.offt_hip_set_output_scale(po, 2.0 ** -14)
    for _ in range(3):
        api.offt_3d_execute(po, dev.data_ptr(), dev.data_ptr())
    print("device ms of the last transform:", 1e3 * L.offt_hip_last_device_seconds(po))
    api.offt_3d_fin(po)
    L.offt_hip_finalize_world()


def summarize(d):
    rows = []
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"]
            kind = "FFT " if "fft_" in name else ("RCCL" if ("nccl" in name.lower() or "rccl" in name.lower()) else None)
            if kind:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?"), name))
    rows.sort()
    # the last transform = everything after the last-but-one gap of > 1 ms ... simpler: take the last third of the launches
    k = len(rows) // 3
    rows = rows[-k:]
    t0 = rows[0][0]
    print("# one transform (the last of three): start us, end us, kind, queue, kernel -- FFT = compute stream, RCCL = comm stream")
    overl = 0
    for i, (s, e, kind, q, name) in enumerate(rows):
        short = name.split("(")[0][:70]
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {kind}  q{q}  {short}")
    fft = [(s, e) for s, e, k_, *_ in rows if k_ == "FFT "]
    rc = [(s, e) for s, e, k_, *_ in rows if k_ == "RCCL"]
    for s, e in fft:
        for s2, e2 in rc:
            overl += max(0, min(e, e2) - max(s, s2))
    tot_f = sum(e - s for s, e in fft)
    tot_r = sum(e - s for s, e in rc)
    span = rows[-1][1] - t0
    print(f"# FFT kernels {tot_f / 1e6:.3f} ms, RCCL kernels {tot_r / 1e6:.3f} ms, both running at once {overl / 1e6:.3f} ms, span {span / 1e6:.3f} ms")
    print("# (on ONE GPU the 'exchange' is a device-to-device copy that shares the HBM with the FFT kernels; the point of the trace is")
    print("#  the ORDER: FFT launches of later tiles / chunks start while earlier tiles' / chunks' exchanges are still running)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--schedule", default="pencil", choices=["pencil", "slab"])
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--comm-streams", type=int, default=0)
    ap.add_argument("--summarize", default=None)
    a = ap.parse_args()
    summarize(a.summarize) if a.summarize else run(a)
