#!/bin/bash
# One GPU-box visit: parity tests, then (only if pytest ended normally) smoke + a short bench.
# usage: scripts/gpu_check.sh <tag> [bench args]
set -u
tag=${1:-run}; shift || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -n 25 gpurun_out/pytest_$tag.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/smoke_$tag.log 2>&1; src=$?
tail -n 3 gpurun_out/smoke_$tag.log; echo "smoke rc=$src"
if [ $src -ne 0 ] && [ $src -ne 1 ]; then exit $src; fi
timeout -k 10 600 python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; brc=$?
cat gpurun_out/bench_$tag.err | tail -n 25; cat gpurun_out/bench_$tag.json; echo "bench rc=$brc"
exit $rc
