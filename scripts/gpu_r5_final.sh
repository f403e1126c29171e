#!/bin/bash
# round 5: records of a build.  usage: scripts/gpu_r5_final.sh <tag> [notests]
#   suite + smoke; headline line (cpu_baseline + flow_err); one-sample shapes with and without the per-launch events; the other BASELINE
#   shapes; real-video loop; self-launched RCCL lines (world 1: the all_gather block); 4-rank gloo rehearsal of the N > 1 line; training
#   line; sustained 2000 steps; steady-state traces of the headline and one-sample shapes; PMC passes of the headline shape
set -u
tag=${1:-r05z}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
  tail -n 3 $o/pytest_$tag.log | cut -c1-200; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
fi
run() { name=$1; shift; timeout -k 10 600 "$@" > $o/${name}_$tag.json 2> $o/${name}_$tag.err || { echo "$name failed"; tail -5 $o/${name}_$tag.err; return 1; }; python3 -c "
import json,sys
for l in open('$o/${name}_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d.get('ms_per_step', d.get('seconds_total')), d.get('value'), d.get('unit'), (d.get('roofline') or {}).get('frac'), (d.get('flow_err') or {}).get('worst'), (d.get('all_gather') or {}).get('exposed_ms'))"; }
run bench python3 bench.py || exit 1
grep -v amdgpu.ids $o/bench_$tag.err | head -17 > $o/bench_${tag}_layers.txt
run driver_like python3 bench.py --gpus 1 --steps 20 --warmup 5
for s in "cfg0 1 256 256" "b1 1 384 512"; do
  set -- $s
  run $1 python3 bench.py --batch $2 --height $3 --width $4 --steps 400 --warmup 50 --cpu-seconds 4 --no-secondary
  grep -v amdgpu.ids $o/$1_$tag.err | head -17 > $o/$1_${tag}_layers.txt
  run $1_noevents python3 bench.py --batch $2 --height $3 --width $4 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary --no-flow-err --no-kernel-events
done
run cfg1c6 python3 bench.py --cin 6 --no-cpu-baseline --no-secondary
run cfg2_st python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --st-warp affine
run cfg2 python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-flow-err
run cfg4 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-flow-err
run cfg5 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-flow-err --vgg16
run stream1 python3 bench_stream.py --clips 1
run stream8 python3 bench_stream.py --clips 8
VSTAB_FORCE_DIST=1 run clip_rccl1 python3 bench_clip.py --gpus 1 --frames 64
VSTAB_FORCE_DIST=1 run bench_rccl1 python3 bench.py --gpus 1 --no-cpu-baseline --no-secondary
run bench_gloo4 python3 bench.py --gpus 4 --backend gloo --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --batch 2 --height 256 --width 256
run train python3 bench_train.py --phases
grep -v amdgpu.ids $o/train_$tag.err | tail -69 > $o/train_${tag}_calls.txt
run sustained python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-secondary --no-flow-err
bash scripts/gpu_r5m.sh $tag "cfg1 b1 cfg0" > $o/steady_$tag.log 2>&1 || { tail -5 $o/steady_$tag.log; exit 1; }
grep "steady state\|conv_mfma_kernel<128, 128\|dominant" $o/rocprof_${tag}_cfg1_steady.md | head -6 | cut -c1-250
bash scripts/gpu_profile.sh $tag > $o/profile_$tag.log 2>&1 || { tail -5 $o/profile_$tag.log; exit 1; }
python3 scripts/pmc_summary.py $o/prof_$tag $o/pmc_$tag.json > $o/pmc_$tag.md; grep "pf2_glue_warp\|conv_mfma_kernel<128, 128\|tap_panel\|conv_rowwin" $o/pmc_$tag.md | cut -c1-250
