#!/bin/bash
# round 5, last refresh: suite, then the lines that changed after the r05p records (bench.py's event sampling, the input-assembly kernel)
set -u
tag=${1:-r05s}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
o=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout=600 > $o/pytest_$tag.log 2>&1; rc=$?
tail -n 3 $o/pytest_$tag.log | cut -c1-200; echo "pytest rc=$rc"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1
run() { name=$1; shift; timeout -k 10 600 "$@" > $o/${name}_$tag.json 2> $o/${name}_$tag.err || { echo "$name failed"; tail -5 $o/${name}_$tag.err; return 1; }; python3 -c "
import json
for l in open('$o/${name}_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d.get('ms_per_step', d.get('ms_per_step_async')), d.get('value'), d.get('unit'), (d.get('roofline') or {}).get('frac'), (d.get('flow_err') or {}).get('worst'))"; }
run bench python3 bench.py || exit 1
grep -v amdgpu.ids $o/bench_$tag.err | head -17 > $o/bench_${tag}_layers.txt
run driver_like python3 bench.py --gpus 1 --steps 20 --warmup 5
run cfg0 python3 bench.py --batch 1 --height 256 --width 256 --steps 400 --warmup 50 --cpu-seconds 4 --no-secondary
run b1 python3 bench.py --batch 1 --height 384 --width 512 --steps 400 --warmup 50 --cpu-seconds 4 --no-secondary
run stream1 python3 bench_stream.py --clips 1
run stream8 python3 bench_stream.py --clips 8
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_stream8_$tag -- python3 bench_stream.py --clips 8 --frames 60 --warmup 10 > /dev/null 2>&1
python3 scripts/prof_summary.py $(find $o/prof_stream8_$tag -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats: bench_stream.py --clips 8 --frames 60 --warmup 10 (8 clips in lockstep, net 384x512, 720p frames), build $tag" 20 > $o/rocprof_${tag}_stream8.md
find $o/prof_stream8_$tag -name "*.csv" -size +4M -delete
head -14 $o/rocprof_${tag}_stream8.md | cut -c1-140
