#!/bin/bash
# developer probe: SQ / LDS / wait counters per kernel of one dev_shape.py run (two --pmc passes, 8 SQ slots each)
#   tools/dev_pmc2.sh <tag> <dev_shape.py args ...>      e.g.  tools/dev_pmc2.sh f32_2048 2048,64,2048 f32 0 2
# summary -> gpurun_out/pmc_<tag>/summary.txt (copy into profiles/ to keep it)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 tools/dev_shape.py "$@" > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/dev_shape.py "$@" > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 - "$OUT" "$*" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, re, sys
out, what = sys.argv[1], sys.argv[2]
def short(name):
    m = re.search(r"(fft_panelx?_k)<([\w ()]+?), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+), (\w+), (\w+)", name)
    if m:
        k, t, N, E, r0, r1, r2, cols, inc, outc, split = m.groups()
        return f"{k}<{t},N={N},{'E' if k == 'fft_panel_k' else 'TPL'}={E},{r0}x{r1}x{r2},cols={cols},{'C' if inc == 'true' else 'S'}{'C' if outc == 'true' else 'S'},{'split' if split == 'true' else 'packed'}>"
    return re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:80]
print("# rocprofv3 --pmc (two passes) of: tools/dev_shape.py", what)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for run in ("a", "b"):
    for f in glob.glob(f"{out}/{run}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fft_" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count", "?"), r.get("Scratch_Size", r.get("Private_Segment_Size", "?")))
for k, d in acc.items():
    g, w, lds, vg, ag, sc = meta[k]
    print(f"{k}\n   grid={g} wg={w} lds={lds}B vgpr={vg} agpr={ag} scratch={sc}")
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(m):
        print("   %-24s n=%d mean=%.5g" % (c, len(d[c]), m[c]))
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        print("   -> LDS bank-conflict share of LDS cycles: %.1f %%" % (100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]))
    if "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
        print("   -> VALU-active / busy: %.3f   LDS-active / busy: %.3f" % (m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_BUSY_CYCLES"], m.get("SQ_ACTIVE_INST_LDS", 0) / m["SQ_BUSY_CYCLES"]))
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
        pass
    if m.get("SQ_ACTIVE_INST_ANY") and m.get("SQ_WAIT_ANY"):
        tot = m["SQ_ACTIVE_INST_ANY"] + m["SQ_WAIT_ANY"] + m.get("SQ_WAIT_INST_ANY", 0)
        print("   -> wave time: waiting (waitcnt/barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%" % (100 * m["SQ_WAIT_ANY"] / tot, 100 * m.get("SQ_WAIT_INST_ANY", 0) / tot, 100 * m["SQ_ACTIVE_INST_ANY"] / tot))
PY
find $OUT -name "*.db" -delete
