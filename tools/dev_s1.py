#!/usr/bin/env python3
"""Developer probe: S=1 (x-y-z output, in-place strided y and x passes) with per-axis variants of a dev build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dev_perf
for var in ((1, 1, 1), (4, 4, 1), (7, 7, 1), (0, 0, 1), (4, 1, 1), (1, 4, 1)):
    dev_perf.run(1024, 1, 0, var, reps=4)
for var in ((1, 1, 1), (4, 1, 1), (7, 1, 1)):
    dev_perf.run(1024, 0, 1, var, reps=4)
