#!/usr/bin/env python3
"""Steady-state summary of a rocprofv3 `--kernel-trace` CSV of `bench.py`, for profiles/.

`rocprofv3 --stats` averages EVERY dispatch of the run, and a short profiled run is a ramp (r03p: the eight launches of the
dominant kernel take 1767 us in step 0 and 1424 us in step 12), so its average does not reproduce the bench line's
`roofline.avg_launch_us`, which covers the timed region only.  This script cuts the trace into steps at a kernel that runs
exactly once per step (the glue + warp launch), drops the first SKIP steps (the bench's warm-up plus any ramp you name) and
averages the rest.  Given the bench line of the same run it also prints each MFMA launch's algorithmic flops / time / peak.

usage: prof_steady.py kernel_trace.csv 'title' SKIP_STEPS [bench.json] [marker-substring]
"""
import csv
import json
import sys
from collections import OrderedDict, defaultdict

PEAK_TF = 157.3


def short(name):
    return (name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("vstab::", ""))[:72]


def main():
    path, title, skip = sys.argv[1], sys.argv[2], int(sys.argv[3])
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else None
    # the launch a step ends with: since round 5 the fused tail (pf2_glue_warp_kernel); the round-4 glue + warp launch in older traces
    marker = sys.argv[5] if len(sys.argv) > 5 else None
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps: a step ends with the marker launch
    steps, cur = [], []
    if marker is None:
        marker = "pf2_glue_warp_kernel" if any("pf2_glue_warp_kernel" in r["Kernel_Name"] for r in rows) else "warp3_tile_kernel<true"
    for r in rows:
        cur.append(r)
        if marker in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    if len(steps) <= skip:
        raise SystemExit(f"only {len(steps)} steps in the trace, cannot skip {skip}")
    # the first step also holds the set-up kernels (weight upload, torch fills): never part of the average
    keep = steps[skip:]
    per = OrderedDict()
    for st in keep:
        for r in st:
            d = per.setdefault(r["Kernel_Name"], [0, 0])
            d[0] += 1
            d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = len(keep)
    span = [int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"]) for st in keep]
    busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st) for st in keep]
    print(f"# {title}\n")
    print(f"steady state: steps {skip}..{len(steps) - 1} of {len(steps)} in the trace ({n} averaged; the first {skip} -- warm-up and ramp -- dropped); "
          f"{sum(len(s) for s in keep) / n:.1f} launches per step; kernel time per step {sum(busy) / n / 1e3:.1f} us, "
          f"first launch start to last launch end {sum(span) / n / 1e3:.1f} us\n")
    # the ramp itself, for the record: kernel time of the first steps against the steady average
    ramp = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st if marker not in r["Kernel_Name"] or True) for st in steps[:min(len(steps), 16)]]
    print("kernel time of the first steps of the trace (us): " + " ".join(f"{x / 1e3:.0f}" for x in ramp) + "\n")
    print("| kernel | launches/step | avg us | us/step | % of kernel time |\n|---|---:|---:|---:|---:|")
    tot = sum(v[1] for v in per.values())
    for name, (c, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {c / n:.2f} | {ns / c / 1e3:.2f} | {ns / n / 1e3:.1f} | {100.0 * ns / tot:.2f} |")
    if bench and bench.get("roofline"):
        rf = bench["roofline"]
        dom = rf["kernel"]
        hit = [(k, v) for k, v in per.items() if dom.replace(" ", "") in k.replace(" ", "")]
        print(f"\nbench line of the same run: {bench['ms_per_step']} ms/step, dominant kernel `{dom}` {rf['launches_per_step']} launches/step, "
              f"`roofline.avg_launch_us` {rf['avg_launch_us']} (HIP events in the library), frac {rf['frac']}")
        for k, (c, ns) in hit:
            avg = ns / c / 1e3
            tf = rf["alg_flops_per_launch_avg"] / (avg * 1e-6) / 1e12
            print(f"this trace, steady state: {avg:.2f} us per launch ({c / n:.1f} per step) -> {rf['alg_flops_per_launch_avg'] / 1e9:.2f} GFLOP / {avg:.2f} us = "
                  f"{tf:.1f} TFLOP/s = {tf / PEAK_TF:.4f} of {PEAK_TF}; trace / events = {avg / rf['avg_launch_us']:.4f}")


if __name__ == "__main__":
    main()
