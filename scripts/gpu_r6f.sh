#!/bin/bash
# round 6: split-K = 2 for the small Winograd stages (conv5_1, conv6_1 at B=8 512x512: 128x64 tiles, one workgroup per CU) with the combine folded
# into the inverse transform (-DVSTAB_WINO_SPLIT=2 build): parity of the variant, then interleaved A/B against the product library
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06h}
VSTAB_LIB=tools/libvstab_hip_wsplit2.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=300 -k "every_layer or cfg1 or golden" > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 3 $o/pytest_$tag.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
one() { name=$1; lib=$2; shift 2; env $lib python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
for i in 1 2 3 4; do
  one base VSTAB_X=0 --steps 40 --warmup 10
  one split2 VSTAB_LIB=tools/libvstab_hip_wsplit2.so --steps 40 --warmup 10
done
for n in base split2; do echo "== $n"; grep -A17 "^launch" $o/ab_${tag}_$n.err | grep "conv5_1\|conv6_1\|all conv" | cut -c1-110; done
timeout -k 10 200 env VSTAB_LIB=tools/libvstab_hip_wsplit2.so true
