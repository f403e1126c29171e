#!/bin/bash
# round 4: records of a build: headline line with the CPU baseline, the other BASELINE shapes, one-sample and real-video modes, sharded
# clip and training lines (self-launched, RCCL world 1), steady-state rocprofv3 traces.  usage: scripts/gpu_r4_final.sh <tag> [notests]
set -u
tag=${1:-r04p}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
  tail -n 3 $o/pytest_$tag.log | cut -c1-200; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
fi
run() { name=$1; shift; timeout -k 10 600 "$@" > $o/${name}_$tag.json 2> $o/${name}_$tag.err || { echo "$name failed"; tail -5 $o/${name}_$tag.err; return 1; }; python3 -c "
import json,sys
for l in open('$o/${name}_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d.get('ms_per_step', d.get('seconds_total')), d.get('value'), d.get('unit'), (d.get('roofline') or {}).get('frac'))"; }
run bench python3 bench.py || exit 1
grep -v amdgpu.ids $o/bench_$tag.err | head -17 > $o/bench_${tag}_layers.txt
run cfg0 python3 bench.py --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary
grep -v amdgpu.ids $o/cfg0_$tag.err | head -17 > $o/cfg0_${tag}_layers.txt
run b1 python3 bench.py --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary
grep -v amdgpu.ids $o/b1_$tag.err | head -17 > $o/b1_${tag}_layers.txt
run cfg1c6 python3 bench.py --cin 6 --no-cpu-baseline --no-secondary
run cfg2_st python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --st-warp affine
run cfg2 python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run cfg4 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run cfg5 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --vgg16
run stream1 python3 bench_stream.py --clips 1
run stream8 python3 bench_stream.py --clips 8
VSTAB_FORCE_DIST=1 run clip_rccl1 python3 bench_clip.py --gpus 1 --frames 64
VSTAB_FORCE_DIST=1 run bench_rccl1 python3 bench.py --gpus 1 --no-cpu-baseline --no-secondary
run train python3 bench_train.py --phases
run sustained python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-secondary
# steady-state traces
P=$o/prof_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/cfg1 -- python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $P.cfg1.json 2> $P.cfg1.err || { tail -5 $P.cfg1.err; exit 1; }
python3 scripts/prof_steady.py $(find $P/cfg1 -name '*kernel_trace.csv' | head -1) "rocprofv3 --kernel-trace, steady state: bench.py --steps 60 --warmup 20 (B=8 512x512x27), build $tag" 30 $o/sustained_$tag.json > $o/rocprof_${tag}_steady.md; head -22 $o/rocprof_${tag}_steady.md; tail -2 $o/rocprof_${tag}_steady.md
cp $(find $P/cfg1 -name '*kernel_stats.csv' | head -1) $o/rocprof_${tag}_kernel_stats.csv
for shape in "1 384 512 b1" "1 256 256 cfg0"; do
  set -- $shape
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/$4 -- python3 bench.py --batch $1 --height $2 --width $3 --steps 100 --warmup 50 --no-cpu-baseline --no-secondary --no-kernel-events > $P.$4.json 2> $P.$4.err || { tail -5 $P.$4.err; exit 1; }
  python3 scripts/prof_steady.py $(find $P/$4 -name '*kernel_trace.csv' | head -1) "rocprofv3 --kernel-trace, steady state: bench.py --batch $1 --height $2 --width $3 --steps 100 --warmup 50 --no-kernel-events, build $tag" 70 > $o/rocprof_${tag}_$4_steady.md; head -20 $o/rocprof_${tag}_$4_steady.md
done
find $P -name '*.csv' -size +6M -delete
