#!/bin/bash
# round 6: non-temporal LOADS of M in the Winograd inverse transform of large stages (-DVSTAB_NT_M_LOADS build): interleaved A/B + kernel times
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06k}
one() { name=$1; lib=$2; shift 2; env $lib python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
for i in 1 2 3 4; do
  one base VSTAB_X=0 --steps 40 --warmup 10
  one ntm VSTAB_LIB=tools/libvstab_hip_ntm.so --steps 40 --warmup 10
done
for v in base ntm; do
  if [ $v = ntm ]; then export VSTAB_LIB=tools/libvstab_hip_ntm.so; else unset VSTAB_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$v -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events > $o/prof_${tag}_$v.log 2>&1
  python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$o/prof_${tag}_$v/**/*kernel_trace.csv",recursive=True)[0]
agg=defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size_X"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("== $v", round(sum(sum(v[len(v)//2:]) for v in agg.values())/15/1e3,1), "us kernel time per step")
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    w=v[len(v)//2:]
    if "wino_output" in k[0] or "conv_mfma_kernel<128, 128" in k[0]: print(k, len(v), round(sum(w)/len(w)/1e3,1))
PY
done
find $o -name '*kernel_trace.csv' -size +6M -delete
