#!/bin/bash
# developer probe: HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes) of one dev_perf.py run:  tools/dev_pmc_hbm.sh <N> [f64|f32]
export TMPDIR=/tmp
N=${1:-1000}; P=${2:-f64}
MODE=zyx; [ "$P" = f32 ] && MODE=f32   # dev_perf.py: "zyx" = f64 z-y-x, "f32" = f32 z-y-x and x-y-z
OUT=gpurun_out/pmc_hbm_$N; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/dev_perf.py $N $MODE > $OUT/f.log 2>&1 || { tail -5 $OUT/f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/dev_perf.py $N $MODE > $OUT/w.log 2>&1 || { tail -5 $OUT/w.log; exit 1; }
python3 - <<PY
import csv, glob, collections
esz = 16 if "$P" == "f64" else 8
alg = $N ** 3 * esz
for run, mul in (("f", 2 * 1024), ("w", 1024)):   # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "fft_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]) * mul)
        for c, v in acc.items():
            print("%s: %d launches, mean %.4g B = %.3f x algorithmic (%.4g B)" % (c, len(v), sum(v) / len(v), sum(v) / len(v) / alg, alg))
PY
