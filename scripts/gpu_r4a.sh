#!/bin/bash
# round 4, visit a: the steady-state rocprofv3 pass that reproduces the bench line (review item 1), and the one-sample
# baselines (B=1 384x512, cfg0 B=1 256x256: bench lines + steady-state traces) of the build the round starts from.
# usage: scripts/gpu_r4a.sh <tag>
set -u
tag=${1:-r04a}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
P=gpurun_out/prof_$tag
# 1. headline shape: 20 warm-up + 60 timed steps under the profiler; the summary averages steps 30.. only
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/cfg1 -- python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-secondary > $P.cfg1.json 2> $P.cfg1.err || { tail -5 $P.cfg1.err; exit 1; }
t=$(find $P/cfg1 -name '*kernel_trace.csv' | head -1)
python3 scripts/prof_steady.py $t "rocprofv3 --kernel-trace, steady state: bench.py --steps 60 --warmup 20 (B=8 512x512x27), build $tag" 30 $P.cfg1.json > gpurun_out/rocprof_${tag}_steady.md || exit 1
head -24 gpurun_out/rocprof_${tag}_steady.md; tail -3 gpurun_out/rocprof_${tag}_steady.md
cp $(find $P/cfg1 -name '*kernel_stats.csv' | head -1) gpurun_out/rocprof_${tag}_kernel_stats.csv
# 2. the same run without the profiler (the line the summary must agree with)
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err || { tail -5 gpurun_out/bench_${tag}.err; exit 1; }
grep -v amdgpu.ids gpurun_out/bench_${tag}.err | tail -22; cut -c1-400 gpurun_out/bench_${tag}.json
# 3. one-sample shapes: bench lines, then steady-state traces
for shape in "1 384 512 b1" "1 256 256 cfg0"; do
  set -- $shape
  timeout -k 10 300 python3 bench.py --batch $1 --height $2 --width $3 --steps 400 --warmup 50 --no-cpu-baseline --no-secondary > gpurun_out/bench_${tag}_$4.json 2> gpurun_out/bench_${tag}_$4.err || { tail -5 gpurun_out/bench_${tag}_$4.err; exit 1; }
  grep -v amdgpu.ids gpurun_out/bench_${tag}_$4.err | head -18; cut -c1-200 gpurun_out/bench_${tag}_$4.json
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/$4 -- python3 bench.py --batch $1 --height $2 --width $3 --steps 100 --warmup 50 --no-cpu-baseline --no-secondary --no-kernel-events > $P.$4.json 2> $P.$4.err || { tail -5 $P.$4.err; exit 1; }
  t=$(find $P/$4 -name '*kernel_trace.csv' | head -1)
  python3 scripts/prof_steady.py $t "rocprofv3 --kernel-trace, steady state: bench.py --batch $1 --height $2 --width $3 --steps 100 --warmup 50 --no-kernel-events, build $tag" 70 > gpurun_out/rocprof_${tag}_$4_steady.md || exit 1
  head -24 gpurun_out/rocprof_${tag}_$4_steady.md
done
find $P -name '*.csv' -size +6M -delete
du -sh $P
