#!/bin/bash
# round 5: training step after the tap-table heads + column-tail split: its tests, then the line with the per-call table
set -u
tag=${1:-r05f}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kloop.py tests/test_gpu_training.py tests/test_gpu_train_step.py -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 15 $o/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench_train.py --phases --no-cpu-baseline > $o/train_$tag.json 2> $o/train_$tag.err || { tail -5 $o/train_$tag.err; exit 1; }
cut -c1-300 $o/train_$tag.json; grep -v amdgpu.ids $o/train_$tag.err | tail -75
