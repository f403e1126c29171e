#!/bin/bash
# round 6: non-temporal stores for the V planes of LARGE Winograd input transforms (>= VSTAB_NT_MIN_BYTES): interleaved A/B against a build
# that never uses them (-DVSTAB_NT_MIN_BYTES='(1ll<<60)') + kernel times
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06i}
one() { name=$1; lib=$2; shift 2; env $lib python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'], d['flow_err'] if 'flow_err' in d else '')"; }
for i in 1 2 3 4; do
  one base VSTAB_LIB=tools/libvstab_hip_nont.so --steps 40 --warmup 10
  one nt VSTAB_X=0 --steps 40 --warmup 10
done
one cfg2_base VSTAB_LIB=tools/libvstab_hip_nont.so --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
one cfg2_nt VSTAB_X=0 --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
one cfg2_base VSTAB_LIB=tools/libvstab_hip_nont.so --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
one cfg2_nt VSTAB_X=0 --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
one cfg4_base VSTAB_LIB=tools/libvstab_hip_nont.so --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1
one cfg4_nt VSTAB_X=0 --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1
one b1_base VSTAB_LIB=tools/libvstab_hip_nont.so --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
one b1_nt VSTAB_X=0 --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wdec.py tests/test_gpu_training.py -m gpu -q -x --timeout=300 > $o/pytest_$tag.log 2>&1; tail -n 2 $o/pytest_$tag.log
for v in base nt; do
  if [ $v = base ]; then export VSTAB_LIB=tools/libvstab_hip_nont.so; else unset VSTAB_LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$v -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events > $o/prof_${tag}_$v.log 2>&1
  python3 - <<PY
import csv,glob
from collections import defaultdict
f=glob.glob("$o/prof_${tag}_$v/**/*kernel_trace.csv",recursive=True)[0]
agg=defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size_X"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("== $v")
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    w=v[len(v)//2:]
    if "input" in k[0] or "wino_gemm" in k[0] or "conv_dual_kernel<64" in k[0]: print(k, len(v), round(sum(w)/len(w)/1e3,1))
PY
done
find $o -name '*kernel_trace.csv' -size +6M -delete
