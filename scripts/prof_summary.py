#!/usr/bin/env python3
"""Markdown table of a rocprofv3 `*_kernel_stats.csv` (top N rows), for profiles/.  usage: prof_summary.py stats.csv 'title' [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
print(f"# {sys.argv[2]}\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("vstab::", "")[:70]
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
