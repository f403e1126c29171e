#!/bin/bash
# round 4, visit h: tap tiles dispatched first in the two-problem launch; A/B at the headline shape of flags 0 (two-problem launch only
# while co-resident) vs 4 (always) and the one-sample shapes again (dispatch order changed)
# HISTORICAL: plan flag 4 ('two-problem launch always') existed only in the build this visit measured; the product now dispatches the tap
# tiles first whenever the pair is not co-resident (csrc/conv_mfma.hip, conv_dual_kernel).  Kept as the record of how ab_r04h_* was taken.
set -u
tag=${1:-r04h}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest tests/test_gpu_skinny.py tests/test_gpu_parity.py -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -n 4 gpurun_out/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
: > gpurun_out/ab_${tag}.txt
for r in 1 2 3; do for f in 0 4; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --plan-flags $f > gpurun_out/ab_${tag}_f${f}_cfg1.json 2> gpurun_out/ab_${tag}_f${f}_cfg1.err || { tail -5 gpurun_out/ab_${tag}_f${f}_cfg1.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/ab_${tag}_f${f}_cfg1.json')); print('round $r flags $f cfg1', d['ms_per_step'], 'ms  all-conv', d['roofline']['all_mfma_launches']['ms_per_step'])" | tee -a gpurun_out/ab_${tag}.txt
done; done
for f in 0 4; do echo "== flags $f cfg1"; grep -v amdgpu.ids gpurun_out/ab_${tag}_f${f}_cfg1.err | head -17; done >> gpurun_out/ab_${tag}.txt
bash scripts/gpu_ab_flags.sh ${tag}_b1 "0 2" 1 | grep "^round"
