#!/bin/bash
# round 6: XCD-contiguous block order of the Winograd input transforms (overlapping patches): parity + interleaved A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06f}
timeout -k 10 500 python -m pytest tests/test_gpu_wdec.py tests/test_gpu_parity.py tests/test_gpu_training.py -m gpu -q -x --timeout=300 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 4 $o/pytest_$tag.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
one() { name=$1; lib=$2; shift 2; env $lib python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
OLD=VSTAB_LIB=tools/libvstab_hip_noxcdin.so; NEW=VSTAB_X=0
for i in 1 2 3 4; do
  one cfg1_old $OLD --steps 40 --warmup 10
  one cfg1_new $NEW --steps 40 --warmup 10
done
for i in 1 2; do
  one cfg2_old $OLD --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
  one cfg2_new $NEW --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
  one b1_old $OLD --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
  one b1_new $NEW --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
done
for v in new; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$v -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events > $o/prof_${tag}_$v.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$o/prof_${tag}_new/**/*kernel_trace.csv",recursive=True)[0]
agg=defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size_X"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    w=v[len(v)//2:]
    if "input" in k[0] or "wdec" in k[0]: print(k, len(v), round(sum(w)/len(w)/1e3,1))
PY
done
