#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_<tag>/) into small tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv      per-kernel calls / total / average ns (from --kernel-trace --stats)
  profiles/<tag>_pmc_hbm.csv           per-dispatch FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
  profiles/pmc_traffic.json            HBM bytes per launch of each pass kernel, corrected as
                                       MI355X_MICROARCH.md prescribes for gfx950:
                                       read bytes = 2 * FETCH_SIZE * 1024 (FETCH_SIZE tallies 128-B requests at 64 B),
                                       write bytes = WRITE_SIZE * 1024.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.search(r"fft_panel_k<([\w ()]+?), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+), (\w+)(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?>", name)
    if m:
        t, N, E, r0, r1, r2, cols, inc, outc, split, r2c, keep, _tw4 = m.groups()
        return (f"fft_panel_k<{t},N={N},E={E},radix={r0}x{r1}x{r2},cols={cols},in_contig={inc},out_contig={outc},"
                f"split={split}{',real_in' if r2c == 'true' else ''}{',keep' if keep in ('true', '1') else ''}>")
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:100]


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]  # gpurun merges into gpurun_out/ without deleting earlier runs: keep the latest only


rows = []
for fn in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(fn)):
        rows.append([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    w.writerows(rows)

# per-pass durations from the kernel trace: the three panel launches of one transform are
# dispatched in the order z, y, x (two of them may be the same kernel symbol with different strides)
PASS = ["z", "y", "x"]


def passes_of(disp):
    """pass of every fft_panel_k dispatch (sorted by start).  With the y and x launches alternating over groups of
    z-planes (the default up to 1024-point lines) a transform is 1 z launch + G (y, x) pairs: the y launches are the
    `keep` twin of the contig-in/strided-out kernel, the x launches the contig/contig kernel.  Without it: z, y, x."""
    names = [short(r["Kernel_Name"]) for r in disp]
    if any(",keep>" in n for n in names):
        return ["y" if ",keep>" in n else ("x" if "out_contig=true" in n else "z") for n in names]
    return [PASS[i % 3] for i in range(len(disp))]


trace = {p: [] for p in PASS}
for fn in newest(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
    disp = [r for r in csv.DictReader(open(fn)) if "fft_panel_k" in r["Kernel_Name"]]
    disp.sort(key=lambda r: int(r["Start_Timestamp"]))
    for p, r in zip(passes_of(disp), disp):
        trace[p].append((short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(os.path.join(dst, f"{tag}_pass_durations.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Pass", "Kernel", "Launches", "AverageNs", "MinNs", "MaxNs", "LaunchesPerTransform", "NsPerTransform"])
    ntr = max(1, len(trace["z"]))
    for p in PASS:
        if trace[p]:
            d = [x[1] for x in trace[p]]
            w.writerow([p, trace[p][0][0], len(d), sum(d) / len(d), min(d), max(d), len(d) / ntr, sum(d) / ntr])

pm = {}
out_rows = []
for ctr in ("fetch", "write"):
    for fn in newest(os.path.join(src, f"pmc_{ctr}", "*", "*_counter_collection.csv")):
        disp = [r for r in csv.DictReader(open(fn)) if "fft_panel_k" in r["Kernel_Name"]]
        disp.sort(key=lambda r: int(r["Start_Timestamp"]))
        for p, r in zip(passes_of(disp), disp):
            s = short(r["Kernel_Name"])
            out_rows.append([p, s, r["Dispatch_Id"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                             r["Counter_Name"], r["Counter_Value"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
            pm.setdefault((p, s, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc_hbm.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Pass", "Kernel", "Dispatch_Id", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter", "Value_KB", "DurationNs"])
    w.writerows(out_rows)

traffic = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/{tag}_pmc_hbm.csv",
           "correction": "read = 2*FETCH_SIZE*1024 B (gfx950 counts 128-B requests at 64 B), write = WRITE_SIZE*1024 B"}
for (p, s, cn) in sorted(pm):
    if cn != "FETCH_SIZE":
        continue
    f = pm[(p, s, "FETCH_SIZE")]
    wv = pm.get((p, s, "WRITE_SIZE"), [])
    if not wv:
        continue
    rd = 2.0 * 1024.0 * sum(f) / len(f)
    wr = 1024.0 * sum(wv) / len(wv)
    traffic[f"{p}_pass_kernel"] = s
    traffic[f"{p}_pass_hbm_read_bytes"] = rd
    traffic[f"{p}_pass_hbm_write_bytes"] = wr
    traffic[f"{p}_pass_hbm_bytes_per_launch"] = rd + wr
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(open(os.path.join(dst, f"{tag}_pass_durations.csv")).read())
print(json.dumps(traffic, indent=1))
