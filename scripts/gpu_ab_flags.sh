#!/bin/bash
# interleaved A/B of plan flags on ONE box: per round and flag set one bench line of each shape (per-launch tables kept for the last round)
# usage: scripts/gpu_ab_flags.sh <tag> "<flag sets>" [rounds]      e.g. scripts/gpu_ab_flags.sh r04c "0 1 3" 2
set -u
tag=$1; sets=$2; rounds=${3:-2}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
out=gpurun_out/ab_${tag}.txt; : > $out
for r in $(seq 1 $rounds); do
  for f in $sets; do
    for shape in "1 384 512 b1" "1 256 256 cfg0"; do
      set -- $shape
      timeout -k 10 200 python3 bench.py --batch $1 --height $2 --width $3 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary --plan-flags $f > gpurun_out/ab_${tag}_f${f}_$4.json 2> gpurun_out/ab_${tag}_f${f}_$4.err || { tail -5 gpurun_out/ab_${tag}_f${f}_$4.err; exit 1; }
      python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_${tag}_f${f}_$4.json')); print('round $r flags $f $4', d['ms_per_step'], 'ms  all-conv', d['roofline']['all_mfma_launches']['ms_per_step'])" | tee -a $out
    done
  done
done
for f in $sets; do for s in b1 cfg0; do echo "== flags $f $s"; grep -v amdgpu.ids gpurun_out/ab_${tag}_f${f}_$s.err | head -17; done; done >> $out
