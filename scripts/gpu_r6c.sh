#!/bin/bash
# round 6: in-launch split-K reduction on the tiled kernel (plan flag 32 = combine launches): tests, then interleaved A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; tag=${1:-r06e}
timeout -k 10 700 python -m pytest tests/test_gpu_skinny.py tests/test_gpu_parity.py tests/test_gpu_wdec.py tests/test_gpu_clip.py -m gpu -q -x --timeout=300 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 5 $o/pytest_$tag.log | cut -c1-250; [ $rc -eq 0 ] || exit $rc
one() { name=$1; shift; python3 bench.py --no-cpu-baseline --no-secondary --no-flow-err "$@" 2>$o/ab_${tag}_$name.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do
  one b1_off --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events --plan-flags 32
  one b1_on --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-kernel-events
  one cfg0_off --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --no-kernel-events --plan-flags 32
  one cfg0_on --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --no-kernel-events
  one cfg1_off --steps 40 --warmup 10 --plan-flags 32
  one cfg1_on --steps 40 --warmup 10
done
one b1ev_off --batch 1 --height 384 --width 512 --steps 200 --warmup 50 --plan-flags 32
one b1ev_on --batch 1 --height 384 --width 512 --steps 200 --warmup 50
for n in b1ev_off b1ev_on cfg1_off cfg1_on; do echo "== $n"; grep -A17 "^launch" $o/ab_${tag}_$n.err | cut -c1-110; done
