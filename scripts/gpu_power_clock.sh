#!/bin/bash
# round 3: is the chip power-limited in the conv launches?  (a) board power / sclk sampled by rocm-smi during a sustained bench run,
# (b) effective clock per dispatch (GRBM_GUI_ACTIVE / 8 XCDs / duration) on LONG dispatches: B=32 720p, where the quotient is accurate
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
o=gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -40 > $o/smi_idle.txt
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $o/smi_during.txt &
SMI=$!
python3 bench.py --steps 2500 --warmup 20 --no-cpu-baseline --no-secondary --no-kernel-events > $o/power_bench.json 2> $o/power_bench.err
wait $SMI
head -12 $o/smi_idle.txt; echo ...; sed -n '8,30p' $o/smi_during.txt | cut -c1-220
python3 -c "import json;d=json.load(open('$o/power_bench.json'));print('sustained', d['ms_per_step'])"
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $o/clk720 -- python3 bench.py --batch 32 --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-events > $o/clk720.log 2>&1 || { tail -5 $o/clk720.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
cc = glob.glob("gpurun_out/clk720/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/clk720/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for d, c in acc.items():
    if d not in dur or "GRBM_GUI_ACTIVE" not in c: continue
    ns, name = dur[d]
    if ns < 2_000_000: continue          # long dispatches only
    clk = c["GRBM_GUI_ACTIVE"] / 8 / ns * 1e3      # MHz
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    rows.append((name.replace("void vstab::", "").split("(")[0][:60], ns / 1e6, clk, util))
print(f"{'kernel':<62}{'ms':>8}{'MHz':>8}{'mfma_util':>10}")
for r in rows: print(f"{r[0]:<62}{r[1]:>8.2f}{r[2]:>8.0f}{r[3]:>10.3f}")
PY
