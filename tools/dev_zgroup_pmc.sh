#!/bin/bash
# developer probe: HBM bytes read / written per kernel flavour of a 1024^3 f64 transform under the z-group settings given
# in the environment (OFFT_ZGROUP_MIB, OFFT_ZGROUP_STREAMS):  tools/dev_zgroup_pmc.sh <tag>
export TMPDIR=/tmp
TAG=$1; OUT=gpurun_out/pmc_zgroup_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/dev_shape.py 1024,1024,1024 f64 0 2 > $OUT/f.log 2>&1 || { tail -5 $OUT/f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/dev_shape.py 1024,1024,1024 f64 0 2 > $OUT/w.log 2>&1 || { tail -5 $OUT/w.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
alg = 1024 ** 3 * 16
print("# $TAG: OFFT_ZGROUP_MIB=${OFFT_ZGROUP_MIB:-default} OFFT_ZGROUP_STREAMS=${OFFT_ZGROUP_STREAMS:-default}; HBM bytes per TRANSFORM and kernel flavour / algorithmic bytes of one sweep (%.4g B)" % alg)
for run, mul, nm in (("f", 2 * 1024, "read"), ("w", 1024, "written")):   # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % run, recursive=True):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fft_" not in k: continue
            m = re.search(r"(true|false), (true|false), (true|false)(?:, (true|false), (true|false))?>", k)
            fl = ("C" if m.group(1) == "true" else "S") + ("C" if m.group(2) == "true" else "S") + (" keep" if m.group(5) == "true" else "")
            acc[fl] += float(r["Counter_Value"]) * mul; cnt[fl] += 1
        for fl in acc:
            print("%s %-8s: %4d launches, %.3f x one sweep per transform" % (nm, fl, cnt[fl], acc[fl] / 2 / alg))
PY
find $OUT -name "*.db" -delete
