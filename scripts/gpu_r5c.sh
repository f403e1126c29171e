#!/bin/bash
# round 5, third visit: the fused tail (tests, then interleaved A/B of plan flags 4 = two launches vs 0 = one, at the headline and the two
# one-sample shapes) and the weight-stream kernel's new XCD map (both arms carry it: compare flag 4 with profiles/ab_r05b_*).
set -u
tag=${1:-r05c}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_skinny.py tests/test_gpu_bench_contract.py tests/test_gpu_configs.py -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 15 $o/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
: > $o/ab_$tag.txt
for r in 1 2 3; do
  for f in 4 0; do
    for shape in "8 512 512 cfg1 200 20" "1 384 512 b1 400 50" "1 256 256 cfg0 400 50"; do
      set -- $shape
      timeout -k 10 200 python3 bench.py --batch $1 --height $2 --width $3 --steps $5 --warmup $6 --no-cpu-baseline --no-secondary --no-flow-err --plan-flags $f > $o/ab_${tag}_f${f}_$4.json 2> $o/ab_${tag}_f${f}_$4.err || { tail -5 $o/ab_${tag}_f${f}_$4.err; exit 1; }
      python3 -c "
import json
d=json.load(open('$o/ab_${tag}_f${f}_$4.json')); h=d['roofline_hbm']
print('round $r flags $f $4', d['ms_per_step'], 'ms  tail', h['kernel'][:24], h['avg_launch_us'], 'us', [ (r['row'][:14], r['avg_launch_us']) for r in h.get('other_rows', []) if r['row'].startswith('K9')])" | tee -a $o/ab_$tag.txt
    done
  done
done
for f in 4 0; do for s in cfg1 b1 cfg0; do echo "== flags $f $s"; grep -v amdgpu.ids $o/ab_${tag}_f${f}_$s.err | head -17; done; done >> $o/ab_$tag.txt
