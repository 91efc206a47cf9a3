#!/bin/bash
# developer probe: SQ/LDS counters of one dev_perf.py run:  tools/dev_pmc.sh <N> <f64|f32> <tag>
export TMPDIR=/tmp
N=${1:-768}; P=${2:-f64}; TAG=${3:-mix}
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 tools/dev_perf.py $N $P > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -- python3 tools/dev_perf.py $N $P > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for run in ("a", "b"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fft_" not in k: continue
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print("   %-24s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
