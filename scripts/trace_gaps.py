#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 kernel trace: per kernel duration and the idle gap in front of it.
usage: trace_gaps.py <kernel_trace.csv> <launches per step> [step index from the end, default 2]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]); back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
step = rows[len(rows) - back * n: len(rows) - (back - 1) * n]
prev_end = None; busy = 0; gaps = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].replace("void ", "").replace("vstab::", "").split("(")[0][:58]
    print(f"{name:<60}{(e - s) / 1e3:>9.2f} us   gap {gap:>7.2f} us   grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}x{r.get('Grid_Size_Y','')}x{r.get('Grid_Size_Z','')}")
    busy += (e - s) / 1e3; gaps += max(gap, 0); prev_end = e
print(f"step: {len(step)} launches, kernel time {busy:.1f} us, gaps {gaps:.1f} us, span {(int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3:.1f} us")
