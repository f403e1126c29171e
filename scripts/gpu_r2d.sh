#!/bin/bash
# round-2 visit D: full GPU suite (incl. cfg3/cfg4 tests), smoke, marker+kernel trace of one bench run
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --timeout=900 --durations=8 > gpurun_out/r2d_pytest.log 2>&1; rc=$?
tail -n 22 gpurun_out/r2d_pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r2d_smoke.log 2>&1; rc=$?
tail -n 2 gpurun_out/r2d_smoke.log; echo "smoke rc=$rc"
[ $rc -ne 0 ] && exit $rc
rm -rf gpurun_out/r2d_marker; timeout -k 10 300 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d gpurun_out/r2d_marker -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --roctx > gpurun_out/r2d_marker.json 2> gpurun_out/r2d_marker.err; rc=$?
tail -n 3 gpurun_out/r2d_marker.err; ls gpurun_out/r2d_marker/*/ | head; find gpurun_out/r2d_marker -name '*.csv' -size +8M -delete
echo "marker rc=$rc"
exit $rc
