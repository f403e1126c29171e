#!/bin/bash
# round 4, visit f: two launches per refinement level (conv_dual_kernel, combine_predict_up_kernel): suite, then interleaved A/B of
# plan flags 0 (new schedule) / 2 (four launches per level) / 3 (round-3 schedule) on the one-sample shapes and 0 / 2 on the headline shape
set -u
tag=${1:-r04f}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest ${TESTS:-tests} -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -n 12 gpurun_out/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
bash scripts/gpu_ab_flags.sh $tag "0 2 3" 2 | grep "^round"
for r in 1 2; do for f in 0 2; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --plan-flags $f > gpurun_out/ab_${tag}_f${f}_cfg1.json 2> gpurun_out/ab_${tag}_f${f}_cfg1.err || { tail -5 gpurun_out/ab_${tag}_f${f}_cfg1.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/ab_${tag}_f${f}_cfg1.json')); print('round $r flags $f cfg1', d['ms_per_step'], 'ms  all-conv', d['roofline']['all_mfma_launches']['ms_per_step'])" | tee -a gpurun_out/ab_${tag}.txt
done; done
for f in 0 2; do echo "== flags $f cfg1"; grep -v amdgpu.ids gpurun_out/ab_${tag}_f${f}_cfg1.err | head -17; done >> gpurun_out/ab_${tag}.txt
