#!/bin/bash
# Run on the GPU box: start / end of the FFT kernels of one rehearsed transform (do consecutive K1 launches overlap?)
export TMPDIR=/tmp
OUT=gpurun_out/trace_k1_$1; rm -rf $OUT; mkdir -p $OUT; shift
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/rehearse_rank.py "$@" --reps 2 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "fft_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
rows.sort()
rows = rows[-28:]
t0 = rows[0][0]
for s, e, q, g in rows:
    print(f"start {(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f} us  queue/stream {q}  grid {g}")
PY
find $OUT -name "*.db" -delete
