#!/bin/bash
# round 5: the training step under the profiler -- the line with its per-call table, then rocprofv3 --kernel-trace --stats of a short run
set -u
tag=${1:-r05e}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 400 python3 bench_train.py --phases --no-cpu-baseline > $o/train_$tag.json 2> $o/train_$tag.err || { tail -5 $o/train_$tag.err; exit 1; }
cut -c1-300 $o/train_$tag.json; grep -v amdgpu.ids $o/train_$tag.err | tail -75
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_train_$tag -- python3 bench_train.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $o/prof_train_$tag.json 2> $o/prof_train_$tag.err || { tail -5 $o/prof_train_$tag.err; exit 1; }
python3 scripts/prof_summary.py $(find $o/prof_train_$tag -name '*kernel_stats.csv' | head -1) "rocprofv3 --kernel-trace --stats: bench_train.py --steps 6 --warmup 2 (B=8 512x512x27; 8 profiled steps), build $tag" 48 > $o/rocprof_${tag}_train.md
head -56 $o/rocprof_${tag}_train.md | cut -c1-160
find $o/prof_train_$tag -name '*.csv' -size +6M -delete
