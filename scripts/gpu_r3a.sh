#!/bin/bash
# round 3, visit a: sampler tests, sampler bench (+ rocprofv3 stats of the same command), bench lines with the ST rows
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_samplers.py -m gpu -q -x --timeout=500 > gpurun_out/pytest_r3a.log 2>&1; rc=$?
tail -n 5 gpurun_out/pytest_r3a.log
[ $rc -le 1 ] || exit $rc
timeout -k 10 300 python scripts/st_bench.py --out gpurun_out/st_bench_r03b.json 2> gpurun_out/st_bench_r03b.err > /dev/null || exit 1
grep -v amdgpu.ids gpurun_out/st_bench_r03b.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_st_r03b -- python3 scripts/st_bench.py --shapes 32x720x1280 --kinds stab --iters 20 > gpurun_out/prof_st_r03b.log 2>&1 || { tail -5 gpurun_out/prof_st_r03b.log; exit 1; }
find gpurun_out/prof_st_r03b -name '*kernel_stats.csv' | head -1 | xargs -r head -12
find gpurun_out/prof_st_r03b -name '*.csv' -size +4M -delete
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r03b.json 2> gpurun_out/bench_r03b.err || { tail -20 gpurun_out/bench_r03b.err; exit 1; }
tail -n 8 gpurun_out/bench_r03b.err; cat gpurun_out/bench_r03b.json
timeout -k 10 400 python bench.py --no-cpu-baseline --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --st-warp affine > gpurun_out/bench_r03b_cfg2_st.json 2> gpurun_out/bench_r03b_cfg2_st.err || { tail -20 gpurun_out/bench_r03b_cfg2_st.err; exit 1; }
tail -n 6 gpurun_out/bench_r03b_cfg2_st.err; cat gpurun_out/bench_r03b_cfg2_st.json
