#!/bin/bash
# round 4, visit g: decoder tests after the co-residency rule, A/B on the headline shape, the glue + warp cache-state study
set -u
tag=${1:-r04g}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest tests/test_gpu_skinny.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -n 6 gpurun_out/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
: > gpurun_out/ab_${tag}.txt
for r in 1 2; do for f in 0 2; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --plan-flags $f > gpurun_out/ab_${tag}_f${f}_cfg1.json 2> gpurun_out/ab_${tag}_f${f}_cfg1.err || { tail -5 gpurun_out/ab_${tag}_f${f}_cfg1.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/ab_${tag}_f${f}_cfg1.json')); print('round $r flags $f cfg1', d['ms_per_step'], 'ms  all-conv', d['roofline']['all_mfma_launches']['ms_per_step'])" | tee -a gpurun_out/ab_${tag}.txt
done; done
timeout -k 10 300 python3 scripts/warp_bench.py --shapes 8x512x512 --insitu-study --iters 40 > gpurun_out/warp_insitu_study_$tag.json 2> gpurun_out/warp_insitu_study_$tag.err; grep -v amdgpu.ids gpurun_out/warp_insitu_study_$tag.err
timeout -k 10 300 python3 scripts/warp_bench.py --shapes 16x1080x1920 --insitu-study --iters 20 > gpurun_out/warp_insitu_study_1080_$tag.json 2> gpurun_out/warp_insitu_study_1080_$tag.err; grep -v amdgpu.ids gpurun_out/warp_insitu_study_1080_$tag.err
