#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/gpu_profile.sh into one markdown summary
(kernel-trace stats + PMC counters per kernel), for profiles/.
usage: scripts/pmc_summary.py gpurun_out/prof_<tag> > profiles/<name>.md"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("vstab::", "")
    return name.split("(")[0]


def main(root):
    print(f"# rocprofv3 summary of `{os.path.basename(root)}`\n")
    print("Command profiled: `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline` "
          "(stats pass; 13 steps incl. warm-up) and `--steps 3 --warmup 1 --no-kernel-events` (PMC passes, one counter set per run).\n")
    st = glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        print("## `rocprofv3 --kernel-trace --stats` (per kernel)\n")
        print("| kernel | calls | total ms | avg us | % | min us | max us |")
        print("|---|---:|---:|---:|---:|---:|---:|")
        for r in csv.DictReader(open(st[0])):
            print(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | "
                  f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} | {int(r['MinNs'])/1e3:.1f} | {int(r['MaxNs'])/1e3:.1f} |")
        print()
    tr = glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        agg = defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("## per launch shape (kernel, grid) from the kernel trace\n")
        print("| kernel | grid (threads x,y,z) | calls | avg us |")
        print("|---|---|---:|---:|")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print(f"| `{k[0]}` | {k[1]},{k[2]},{k[3]} | {len(v)} | {sum(v)/len(v)/1e3:.1f} |")
        print()
    cnt = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if cnt:
        names = sorted({c for k in cnt.values() for c in k})
        print("## PMC counters, average per launch\n")
        print("FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts a wide coalesced "
              "(16 B/lane) read at half its bytes (MI355X_MICROARCH.md, HBM section) -- `hbm_read_MB_corrected` doubles it; "
              "dword-wide access patterns are uncalibrated.\n")
        print("`mfma_util` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs): fraction of all SIMD-cycles "
              "of the launch in which the matrix pipe was busy.\n")
        print("| kernel | " + " | ".join(names) + " | hbm_read_MB_corrected | hbm_write_MB | mfma_util |")
        print("|---|" + "---:|" * (len(names) + 3))
        js = {}
        for k, d in sorted(cnt.items()):
            avg = {c: sum(v) / len(v) for c, v in d.items()}
            rd = avg.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6
            wr = avg.get("WRITE_SIZE", 0) * 1024 / 1e6
            mf = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024) if avg.get("GRBM_GUI_ACTIVE") else float("nan")
            print(f"| `{k}` | " + " | ".join(f"{avg.get(c, float('nan')):.4g}" for c in names) + f" | {rd:.1f} | {wr:.1f} | {mf:.3f} |")
            js[k] = {"fetch_kib": avg.get("FETCH_SIZE"), "write_kib": avg.get("WRITE_SIZE"),
                     "hbm_bytes_per_launch_corrected": (avg.get("FETCH_SIZE", 0) * 2 + avg.get("WRITE_SIZE", 0)) * 1024,
                     "mfma_util": mf}
        print()
        if len(sys.argv) > 2:
            import json
            json.dump(js, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
