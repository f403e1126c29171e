#!/bin/bash
# round 5: steady-state rocprofv3 traces of the one-sample shapes (and optionally the headline shape) with the round-5 build
set -u
tag=${1:-r05m}; shapes=${2:-"b1 cfg0"}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out; P=$o/prof_$tag
for s in $shapes; do
  case $s in
    b1) a="--batch 1 --height 384 --width 512 --steps 100 --warmup 50 --no-kernel-events"; skip=70;;
    cfg0) a="--batch 1 --height 256 --width 256 --steps 100 --warmup 50 --no-kernel-events"; skip=70;;
    cfg1) a="--steps 60 --warmup 20"; skip=30;;
  esac
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$s -- python3 bench.py $a --no-cpu-baseline --no-secondary --no-flow-err > $P.$s.json 2> $P.$s.err || { tail -5 $P.$s.err; exit 1; }
  python3 scripts/prof_steady.py $(find $P/$s -name '*kernel_trace.csv' | head -1) "rocprofv3 --kernel-trace, steady state: bench.py $a, build $tag" $skip $([ $s = cfg1 ] && echo $P.$s.json) > $o/rocprof_${tag}_${s}_steady.md
  head -34 $o/rocprof_${tag}_${s}_steady.md | cut -c1-170; tail -3 $o/rocprof_${tag}_${s}_steady.md | cut -c1-250
  cut -c1-200 $P.$s.json
done
find $P -name '*.csv' -size +6M -delete
