#!/bin/bash
# round 6: the Winograd transposed-convolution form after merging its inverse transform into predict_up's launch: tests, then interleaved
# A/B (plan flag 8 = direct form) at the headline shape and at the 720p / 1080p batch shapes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06b}
timeout -k 10 600 python -m pytest tests/test_gpu_wdec.py tests/test_gpu_parity.py tests/test_gpu_skinny.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x --timeout=300 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 5 $o/pytest_$tag.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
one() { name=$1; shift; python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
for i in 1 2 3 4; do
  one off --steps 40 --warmup 10 --plan-flags 8
  one on --steps 40 --warmup 10
done
for i in 1 2; do
  one cfg2_off --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --plan-flags 8
  one cfg2_on --batch 32 --height 720 --width 1280 --steps 5 --warmup 2
  one cfg4_off --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --plan-flags 8
  one cfg4_on --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1
done
one b16_off --batch 16 --steps 20 --warmup 5 --plan-flags 8
one b16_on --batch 16 --steps 20 --warmup 5
one p720x4_off --batch 4 --height 720 --width 1280 --steps 20 --warmup 5 --plan-flags 8
one p720x4_on --batch 4 --height 720 --width 1280 --steps 20 --warmup 5
one p1080x2_off --batch 2 --height 1080 --width 1920 --steps 20 --warmup 5 --plan-flags 8
one p1080x2_on --batch 2 --height 1080 --width 1920 --steps 20 --warmup 5
for n in off on cfg2_off cfg2_on cfg4_off cfg4_on; do echo "== $n"; grep -A17 "^launch" $o/ab_${tag}_$n.err | grep "deconv\|all conv" | cut -c1-110; done
