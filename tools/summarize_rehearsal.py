#!/usr/bin/env python3
"""Join the `offt-pass` lines of a rehearsal (OFFT_LOG_PASSES=1, stderr) with the rocprofv3 kernel trace of the same run:
per kernel family and pass shape -> launches, MEDIAN duration (the first transform of a run -- cold: first touch of the
exchange volumes, code objects, translations -- takes up to twice as long per launch and used to sit in the mean: round 3's
earlier per-kernel tables, 3 + 1 transforms per trace, read 10-15 % low for that reason), mean, algorithmic bytes per launch
(2 * S * elements) and the fraction of 8 TB/s at the median.    tools/summarize_rehearsal.py <log with offt-pass lines> <dir of the rocprofv3 run> <reps>"""
import collections
import csv
import glob
import re
import sys

log, rdir = sys.argv[1], sys.argv[2]
passes = []
for line in open(log, errors="replace"):
    m = re.match(r"offt-pass (\S+) n (\d+) ncols (\d+) nb1 (\d+) nb2 (\d+) prec (\d+) in_contig (\d+) out_contig (\d+) in_split (\d+) out_split (\d+) elems (\d+)", line)
    if m:
        passes.append((m.group(1),) + tuple(int(x) for x in m.groups()[1:]))
disp = []
for fn in glob.glob(rdir + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "fft_" in r["Kernel_Name"]:
            disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
disp.sort()
if len(disp) != len(passes):
    print(f"# warning: {len(passes)} logged passes vs {len(disp)} traced fft kernels; matching the common prefix")
acc = collections.OrderedDict()
for p, d in zip(passes, disp):
    fam, n, ncols, nb1, nb2, prec, inc, outc, isp, osp, elems = p
    key = (fam, n, "CS"[0] if inc else "S", "C" if outc else "S", ncols, nb1, nb2, isp, osp, prec)
    acc.setdefault(key, []).append((d[1], elems))
print("# kernel family, n, flavour (in/out contiguous), ncols x nb1 x nb2, in_split/out_split -> launches, median us (mean us), alg MB/launch, frac of 8 TB/s at the median")
tot_t = tot_b = 0.0
for key, v in acc.items():
    fam, n, fi, fo, ncols, nb1, nb2, isp, osp, prec = key
    esz = 16 if prec == 0 else 8
    ds = sorted(x[0] for x in v)
    t = ds[len(ds) // 2] * 1e-9
    mean = sum(ds) / len(ds) * 1e-9
    b = 2.0 * esz * v[0][1]
    tot_t += t * len(v)
    tot_b += sum(2.0 * esz * x[1] for x in v)
    print(f"{fam:13s} n={n:5d} {fi}{fo} {ncols:5d} x {nb1:5d} x {nb2:3d} split {isp}/{osp}: {len(v):4d} launches, {t*1e6:9.1f} us ({mean*1e6:.1f}), {b/1e6:9.1f} MB, {b/t/8e12*100:5.1f} %")
print(f"# all fft launches at their medians: {tot_t*1e3:.3f} ms of kernel time for {tot_b/1e9:.2f} GB algorithmic -> {tot_b/tot_t/8e12*100:.1f} % of 8 TB/s")
