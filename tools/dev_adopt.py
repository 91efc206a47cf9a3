#!/usr/bin/env python3
"""Developer helper: compare a sweep log with the product library's log (same box) and list the shapes worth adopting.
  tools/dev_adopt.py <prod.log> <sweep.log> [min gain, default 0.03]"""
import re
import sys

pat = re.compile(r"N=(\d+) f64 v(\d+) .*radix=(\d+)x(\d+)x(\d+) threads/line=(\d+) .*cols=(\d+) .*\[z/y/x ([\d.]+)/([\d.]+)/([\d.]+) ms\]")


def load(fn):
    d = {}
    for line in open(fn):
        m = pat.match(line)
        if m:
            N, v, r0, r1, r2, tpl, cols, z, y, x = m.groups()
            d.setdefault(int(N), []).append(dict(v=int(v), shape=(int(tpl), int(r0), int(r1), int(r2), int(cols)), z=float(z), y=float(y), x=float(x)))
    return d


prod, sweep = load(sys.argv[1]), load(sys.argv[2])
gain = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
for N in sorted(sweep):
    if N not in prod:
        continue
    p0 = [e for e in prod[N] if e["v"] == 0][0]
    slab = p0["y"] > 3 * max(p0["z"], p0["x"]) or p0["y"] < 0.3 * min(p0["z"], p0["x"])  # N x 256 x N grids: y is another length
    strided = (lambda e: e["z"] if slab else e["z"] + e["y"])
    best = min(sweep[N], key=strided)
    if strided(best) < (1 - gain) * strided(p0):
        tpl, r0, r1, r2, cols = best["shape"]
        print(f"N={N}: strided passes {strided(p0):.3f} -> {strided(best):.3f} ms ({(1 - strided(best) / strided(p0)) * 100:.1f} %), x {p0['x']:.3f} -> {best['x']:.3f}: "
              f"reg_variantx<double, {N}, {tpl}, {r0}, {r1}, {r2}, {cols}, true>   (now {p0['shape']})")
