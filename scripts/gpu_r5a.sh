#!/bin/bash
# round 5, first visit: -m gpu suite, smoke, the headline line WITH the cpu_baseline + flow_err blocks, the one-sample shapes, and the
# parity-headroom table (scripts/flow_err_margin.py).   usage: scripts/gpu_r5a.sh <tag> [notests]
set -u
tag=${1:-r05a}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
rc=0
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 1000 python -m pytest ${TESTS:-tests} -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
  tail -n 25 $o/pytest_$tag.log | cut -c1-400; echo "pytest rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
  timeout -k 10 300 python __graft_entry__.py smoke > $o/smoke_$tag.log 2>&1; src=$?
  tail -n 2 $o/smoke_$tag.log; echo "smoke rc=$src"
  if [ $src -ne 0 ] && [ $src -ne 1 ]; then exit $src; fi
fi
timeout -k 10 400 python3 bench.py > $o/bench_$tag.json 2> $o/bench_$tag.err || { tail -5 $o/bench_$tag.err; exit 1; }
grep -v amdgpu.ids $o/bench_$tag.err | head -17; python3 -c "
import json
d=json.loads([l for l in open('$o/bench_$tag.json') if l.startswith('{')][0])
print('bench', d['ms_per_step'], d['value'], d['roofline']['frac']); print('flow_err', json.dumps(d.get('flow_err'))); print('cpu', d.get('cpu_baseline'))"
for shape in "1 384 512 b1 400 50" "1 256 256 cfg0 400 50"; do
  set -- $shape
  timeout -k 10 300 python3 bench.py --batch $1 --height $2 --width $3 --steps $5 --warmup $6 --cpu-seconds 4 --no-secondary > $o/bench_${tag}_$4.json 2> $o/bench_${tag}_$4.err || { tail -5 $o/bench_${tag}_$4.err; exit 1; }
  grep -v amdgpu.ids $o/bench_${tag}_$4.err | head -17; cut -c1-230 $o/bench_${tag}_$4.json; echo
done
timeout -k 10 900 python3 scripts/flow_err_margin.py --out $o/flow_err_margin_$tag.md --json $o/flow_err_margin_$tag.json > $o/flow_err_margin_$tag.log 2>&1 || { tail -20 $o/flow_err_margin_$tag.log; exit 1; }
grep -c "" $o/flow_err_margin_$tag.md; tail -12 $o/flow_err_margin_$tag.md | cut -c1-300
exit $rc
