#!/usr/bin/env python3
"""Summarise build/kernel_resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, sys
import glob
files = [sys.argv[1]] if len(sys.argv) > 1 and sys.argv[1] != "-" else sorted(glob.glob("build/*.resource_usage.txt"))
txt = "".join(open(f).read() for f in files)
minN = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    m = re.search(r"fft_panel(x?)_kI(\w)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)", name)
    if not m:
        continue
    x, t, N, E, r0, r1, r2, cols, inc, outc, split = m.groups()
    if x:
        E = "tpl" + E
    if int(N) < minN:
        continue
    def g(k):
        mm = re.search(k + r": (\d+)", b)
        return mm.group(1) if mm else "?"
    print(f"{t} N={N} E={E} {r0}x{r1}x{r2} cols={cols} inc={inc} outc={outc} split={split}: VGPR", g(" VGPRs"), "AGPR", g("AGPRs"),
          "spill", g("VGPRs Spill"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "occ", g(r"Occupancy \[waves/SIMD\]"))
