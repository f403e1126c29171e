#!/bin/bash
# round-2 visit C: parity of the productised warp, HBM micro-bench, bench at cfg1 and at B=16 1080p
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_clip.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout=300 > gpurun_out/r2c_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/r2c_tests.log; echo "tests rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/warp_bench.py > gpurun_out/r2c_warp_bench.json 2> gpurun_out/r2c_warp_bench.err; rc=$?
cat gpurun_out/r2c_warp_bench.err | grep -v amdgpu.ids | tail -n 40; echo "warp bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; rc=$?
tail -n 20 gpurun_out/r2c_bench.err; cat gpurun_out/r2c_bench.json; echo "bench rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --batch 16 --height 1080 --width 1920 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2c_bench_1080.json 2> gpurun_out/r2c_bench_1080.err; rc=$?
tail -n 3 gpurun_out/r2c_bench_1080.err; cat gpurun_out/r2c_bench_1080.json; echo "bench1080 rc=$rc"
exit $rc
