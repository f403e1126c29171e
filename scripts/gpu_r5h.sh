#!/bin/bash
# round 5: the 8-bit fused tail in the clip step (tests, stream benches) and the full suite
set -u
tag=${1:-r05h}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 12 $o/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
for c in 1 8; do
  timeout -k 10 300 python3 bench_stream.py --clips $c > $o/stream${c}_$tag.json 2> $o/stream${c}_$tag.err || { tail -5 $o/stream${c}_$tag.err; exit 1; }
  cut -c1-420 $o/stream${c}_$tag.json
done
