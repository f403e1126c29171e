#!/bin/bash
# HBM-side kernels under rocprofv3: per-kernel durations of scripts/warp_bench.py (usage: gpu_warp_prof.sh <tag> [warp_bench args])
set -u
tag=${1:-run}; shift || true
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/warpprof_$tag
rm -rf $out; mkdir -p $out
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/warp_bench.py "$@" > $out/bench.json 2> $out/bench.err; rc=$?
tail -n 40 $out/bench.err
f=$(find $out/stats -name '*kernel_stats*.csv' | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -12 $out/kernel_stats.csv | cut -c1-200
find $out -name '*.csv' -size +8M -delete
exit $rc
