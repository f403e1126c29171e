#!/bin/bash
# round 4, visit j: chained combines (conv_chain_kernel).  The new tests first, under a short time limit of their own (a hand-off bug must
# not sit on the GPU); only then the suite and the A/B.
# HISTORICAL: needs tools/chain_combine_r04.diff applied (plan flag 4 = no chained combines, 7 = the round-3 schedule in that build); kept as the
# record of how profiles/ab_r04j_chained_combines_negative.txt was taken.
set -u
tag=${1:-r04j}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 240 python -m pytest tests/test_gpu_skinny.py -m gpu -q -x --timeout=200 -k "chained" > gpurun_out/pytest_${tag}_chain.log 2>&1; rc=$?
tail -n 15 gpurun_out/pytest_${tag}_chain.log | cut -c1-300; echo "chain tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -n 4 gpurun_out/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_ab_flags.sh $tag "0 4 7" 2 | grep "^round"
for r in 1 2; do for f in 0 4; do
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --plan-flags $f > gpurun_out/ab_${tag}_f${f}_cfg1.json 2> gpurun_out/ab_${tag}_f${f}_cfg1.err || { tail -5 gpurun_out/ab_${tag}_f${f}_cfg1.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/ab_${tag}_f${f}_cfg1.json')); print('round $r flags $f cfg1', d['ms_per_step'], 'ms  all-conv', d['roofline']['all_mfma_launches']['ms_per_step'])" | tee -a gpurun_out/ab_${tag}.txt
done; done
