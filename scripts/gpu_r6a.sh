#!/bin/bash
# round 6: interleaved A/B of the Winograd F(2x2,2x2) transposed-convolution variants at the headline shape (one box visit)
#   off = plan flag 8 (direct form everywhere), on = default plan (deconv3), t64 = 64x128 tiles for the 9-position GEMM,
#   dec4 = threshold lowered so that deconv4 takes the form too
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06a}
one() { name=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err --steps 40 --warmup 10 $EXTRA 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'])"; }
for i in 1 2 3; do
  EXTRA="--plan-flags 8" one off VSTAB_X=0
  EXTRA="" one on VSTAB_X=0
  EXTRA="" one t64 VSTAB_LIB=tools/libvstab_hip_wdec64.so
  EXTRA="" one dec4 VSTAB_LIB=tools/libvstab_hip_wdec4.so
done
for n in off on t64 dec4; do echo "== $n"; grep -A17 "^launch" $o/ab_${tag}_$n.err | grep "deconv\|all conv" | cut -c1-110; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_on -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events > $o/prof_${tag}_on.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$o/prof_${tag}_on/**/*kernel_trace.csv",recursive=True)[0]
agg=defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].split("(")[0][:90], r.get("Grid_Size_X"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    w=v[len(v)//2:]
    print(k, len(v), round(sum(w)/len(w)/1e3,1))
PY
find $o/prof_${tag}_on -name '*.csv' -size +6M -delete
