#!/bin/bash
# One round-4 GPU-box visit: the -m gpu suite (or the files named in TESTS), smoke, then bench lines of the headline shape and the two
# one-sample shapes with their per-launch tables.   usage: [TESTS="tests/a.py tests/b.py"] scripts/gpu_visit.sh <tag> [extra bench args]
set -u
tag=${1:-run}; shift || true
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest ${TESTS:-tests} -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -n 25 gpurun_out/pytest_$tag.log | cut -c1-400; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke_$tag.log 2>&1; src=$?
tail -n 2 gpurun_out/smoke_$tag.log; echo "smoke rc=$src"
if [ $src -ne 0 ] && [ $src -ne 1 ]; then exit $src; fi
for shape in "8 512 512 cfg1 200 20" "1 384 512 b1 400 50" "1 256 256 cfg0 400 50"; do
  set -- $shape
  timeout -k 10 300 python3 bench.py --batch $1 --height $2 --width $3 --steps $5 --warmup $6 --no-cpu-baseline --no-secondary > gpurun_out/bench_${tag}_$4.json 2> gpurun_out/bench_${tag}_$4.err || { tail -5 gpurun_out/bench_${tag}_$4.err; exit 1; }
  grep -v amdgpu.ids gpurun_out/bench_${tag}_$4.err | head -17; cut -c1-230 gpurun_out/bench_${tag}_$4.json; echo
done
exit $rc
