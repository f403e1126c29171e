#!/bin/bash
# round 6: records of a build, in parts that each fit one gpurun call (<= 1200 s).  usage: scripts/gpu_r6_final.sh <tag> <part>
#   part tests   : the -m gpu suite + smoke
#   part lines   : headline line (cpu_baseline + flow_err), driver-style 20-step line, one-sample shapes with and without per-launch events,
#                  the other BASELINE shapes, clip driver, world-1 RCCL lines, training line, sustained run
#   part traces  : steady-state rocprofv3 traces of the headline and one-sample shapes
#   part pmc     : PMC passes of the headline shape (cfg1) -- separate --pmc runs, scripts/gpu_profile.sh
#   part pmcb1   : the same passes for ONE 384x512 sample (the weight-stream kernel's bytes per launch)
set -u
tag=${1:-r06z}; part=${2:-lines}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
run() { name=$1; shift; timeout -k 10 600 "$@" > $o/${name}_$tag.json 2> $o/${name}_$tag.err || { echo "$name failed"; tail -5 $o/${name}_$tag.err; return 1; }; python3 -c "
import json,sys
for l in open('$o/${name}_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d.get('ms_per_step', d.get('seconds_total')), d.get('value'), d.get('unit'), (d.get('roofline') or {}).get('frac'), (d.get('flow_err') or {}).get('worst'), (d.get('all_gather') or {}).get('exposed_ms'))"; }
case $part in
tests)
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
  tail -n 3 $o/pytest_$tag.log | cut -c1-200; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
  ;;
lines)
  run bench python3 bench.py || exit 1
  grep -v amdgpu.ids $o/bench_$tag.err | head -17 > $o/bench_${tag}_cfg1_layers.txt
  run driver_like python3 bench.py --gpus 1 --steps 20 --warmup 5
  for s in "cfg0 1 256 256" "b1 1 384 512"; do
    set -- $s
    run $1 python3 bench.py --batch $2 --height $3 --width $4 --steps 400 --warmup 50 --cpu-seconds 4 --no-secondary
    grep -v amdgpu.ids $o/$1_$tag.err | head -17 > $o/bench_${tag}_$1_layers.txt
    run $1_noevents python3 bench.py --batch $2 --height $3 --width $4 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events
  done
  run cfg1c6 python3 bench.py --cin 6 --no-cpu-baseline --no-secondary
  run cfg2 python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-flow-err
  run cfg4 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-flow-err
  run cfg5 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-flow-err --vgg16
  run stream1 python3 bench_stream.py --clips 1
  run stream8 python3 bench_stream.py --clips 8
  VSTAB_FORCE_DIST=1 run bench_rccl1 python3 bench.py --gpus 1 --no-cpu-baseline --no-secondary
  run train python3 bench_train.py --phases
  grep -v amdgpu.ids $o/train_$tag.err | tail -69 > $o/train_${tag}_calls.txt
  run sustained python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-secondary --no-flow-err
  ;;
traces)
  bash scripts/gpu_r5m.sh $tag "cfg1 b1 cfg0" > $o/steady_$tag.log 2>&1 || { tail -5 $o/steady_$tag.log; exit 1; }
  grep "steady state\|conv_mfma_kernel<128, 128\|dominant" $o/rocprof_${tag}_cfg1_steady.md | head -6 | cut -c1-250
  ;;
pmc)
  bash scripts/gpu_profile.sh $tag > $o/profile_$tag.log 2>&1 || { tail -5 $o/profile_$tag.log; exit 1; }
  python3 scripts/pmc_summary.py $o/prof_$tag $o/pmc_$tag.json > $o/pmc_$tag.md; grep "pf2_glue_warp\|conv_mfma_kernel<128, 128\|tap_panel\|conv_rowwin" $o/pmc_$tag.md | cut -c1-250
  ;;
pmcb1)
  EXTRA="--batch 1 --height 384 --width 512" bash scripts/gpu_profile.sh ${tag}_b1 > $o/profile_${tag}_b1.log 2>&1 || { tail -5 $o/profile_${tag}_b1.log; exit 1; }
  python3 scripts/pmc_summary.py $o/prof_${tag}_b1 $o/pmcb1_$tag.json > $o/pmcb1_$tag.md; grep "conv_skinny\|conv_mfma_kernel<64, 128\|conv_rowwin" $o/pmcb1_$tag.md | cut -c1-250
  ;;
esac
