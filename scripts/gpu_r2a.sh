#!/bin/bash
# round-2 visit A: new warp kernel parity + HBM micro-bench, then full suite, then self-launched RCCL rehearsal
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 -k "warp or glue" > gpurun_out/r2a_warp_tests.log 2>&1; rc=$?
tail -n 8 gpurun_out/r2a_warp_tests.log; echo "warp tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/warp_bench.py > gpurun_out/r2a_warp_bench.json 2> gpurun_out/r2a_warp_bench.err; rc=$?
cat gpurun_out/r2a_warp_bench.err | tail -n 20; echo "warp bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/r2a_pytest.log 2>&1; rc=$?
tail -n 8 gpurun_out/r2a_pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
VSTAB_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a_bench_dist1.json 2> gpurun_out/r2a_bench_dist1.err; rc=$?
tail -n 22 gpurun_out/r2a_bench_dist1.err; cat gpurun_out/r2a_bench_dist1.json; echo "bench dist rc=$rc"
[ $rc -ne 0 ] && exit $rc
VSTAB_FORCE_DIST=1 timeout -k 10 600 python bench_clip.py --gpus 1 --frames 64 > gpurun_out/r2a_clip_dist1.json 2> gpurun_out/r2a_clip_dist1.err; rc=$?
tail -n 5 gpurun_out/r2a_clip_dist1.err; cat gpurun_out/r2a_clip_dist1.json; echo "clip dist rc=$rc"
exit $rc
