#!/usr/bin/env python3
"""Per-kernel means of the counter passes tools/pmc_run.sh collected (rocprofv3 --pmc, csv), with the ratios that say
what a kernel waits for.   tools/summarize_pmc.py <dir with a/ b/ c/> "<what was run>"
Kernels are keyed by short name AND grid size, so that K1 / K2 / K3 of a rehearsal (same template, different shapes) stay apart."""
import collections
import csv
import glob
import re
import sys

out, what = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    m = re.search(r"(fft_panelx?_k)<([\w ()]+?), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+), (\w+)(?:, (\w+), (\w+))?(?:, (\w+))?", name)
    if m:
        k, t, N, E, r0, r1, r2, cols, inc, outc, split, r2c, keep, _tw4 = m.groups()
        t = {"HIP_vector_type<float, 2>": "f32x2"}.get(t, t)
        return (f"{k}<{t},N={N},{'E' if k == 'fft_panel_k' else 'TPL'}={E},{r0}x{r1}x{r2},cols={cols},"
                f"{'C' if inc == 'true' else 'S'}{'C' if outc == 'true' else 'S'},{'split' if split == 'true' else 'packed'}"
                f"{',keep' if keep == 'true' else ''}>")
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:90]


print("# rocprofv3 --pmc (separate passes, no trace) of:", what)
acc = collections.OrderedDict()
meta = {}
for run in ("a", "b", "c"):
    for f in glob.glob(f"{out}/{run}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fft_" not in r["Kernel_Name"]:
                continue
            k = (short(r["Kernel_Name"]), r["Grid_Size"])
            acc.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count", "?"),
                       r.get("Scratch_Size", r.get("Private_Segment_Size", "?")))
for k, d in acc.items():
    w, lds, vg, ag, sc = meta[k]
    wgs = int(k[1]) // max(int(w), 1)
    print(f"{k[0]}\n   grid={k[1]} ({wgs} workgroups = {wgs / 256:.2f} per CU) wg={w} lds={lds}B vgpr={vg} agpr={ag} scratch={sc}")
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        print("   %-32s n=%d mean=%.5g" % (c, len(d[c]), m[c]))
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print("   -> LDS bank-conflict share of LDS cycles: %.1f %%" % (100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]))
    if m.get("SQ_BUSY_CYCLES"):
        print("   -> VALU-active / busy: %.3f" % (m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_BUSY_CYCLES"]))
    if m.get("SQ_ACTIVE_INST_ANY") and m.get("SQ_WAIT_ANY"):
        tot = m["SQ_ACTIVE_INST_ANY"] + m["SQ_WAIT_ANY"] + m.get("SQ_WAIT_INST_ANY", 0)
        print("   -> wave time: waiting (waitcnt/barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%"
              % (100 * m["SQ_WAIT_ANY"] / tot, 100 * m.get("SQ_WAIT_INST_ANY", 0) / tot, 100 * m["SQ_ACTIVE_INST_ANY"] / tot))
    if m.get("SQ_WAVES") and m.get("SQ_WAVE_CYCLES") and m.get("SQ_BUSY_CYCLES"):
        # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES (both summed over SEs/XCDs the same way) = average waves resident while busy
        print("   -> waves launched %.0f; wave-cycles / busy-cycles = %.1f" % (m["SQ_WAVES"], m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"]))
    if m.get("TCP_UTCL1_REQUEST_sum"):
        print("   -> UTCL1 translation misses / requests: %.4f" % (m.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / m["TCP_UTCL1_REQUEST_sum"]))
    if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
        print("   -> L2 hit rate: %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
