#!/bin/bash
# round 3: records of the final build: headline line with the CPU baseline, sustained 2000 steps, the other BASELINE
# shapes, latency and real-video modes, sharded clip and training lines (self-launched, RCCL world 1), sampler bench, rocprofv3 stats + PMC
set -u
tag=${1:-r03p}
mkdir -p gpurun_out; export TMPDIR=/tmp
o=gpurun_out
run() { name=$1; shift; timeout -k 10 600 "$@" > $o/${name}_$tag.json 2> $o/${name}_$tag.err || { echo "$name failed"; tail -5 $o/${name}_$tag.err; return 1; }; python3 -c "
import json,sys
for l in open('$o/${name}_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d.get('ms_per_step', d.get('seconds_total')), d.get('value'), d.get('unit'))"; }
run bench python3 bench.py || exit 1
run sustained python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-secondary || exit 1
run cfg0 python3 bench.py --batch 1 --height 256 --width 256 --steps 50 --warmup 10 --no-cpu-baseline --no-secondary
run cfg1c6 python3 bench.py --cin 6 --no-cpu-baseline --no-secondary
run cfg2_st python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --st-warp affine
run cfg2 python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
run cfg4 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
run cfg5 python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --vgg16
run b1 python3 bench.py --batch 1 --height 384 --width 512 --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-kernel-events
run stream1 python3 bench_stream.py --clips 1
run stream8 python3 bench_stream.py --clips 8
VSTAB_FORCE_DIST=1 run clip_rccl1 python3 bench_clip.py --gpus 1 --frames 64
VSTAB_FORCE_DIST=1 run bench_rccl1 python3 bench.py --gpus 1 --no-cpu-baseline
run train python3 bench_train.py --phases
VSTAB_FORCE_DIST=1 run train_rccl1 python3 bench_train.py --gpus 1
timeout -k 10 300 python3 scripts/st_bench.py --out $o/st_bench_$tag.json 2> $o/st_bench_$tag.err > /dev/null; grep "32x720x1280 stab" $o/st_bench_$tag.err
timeout -k 10 300 python3 scripts/warp_bench.py > $o/warp_bench_$tag.json 2> $o/warp_bench_$tag.err; grep "8x512x512\|16x1080x1920" $o/warp_bench_$tag.err | grep fused | head -4
bash scripts/gpu_profile.sh $tag > $o/profile_$tag.log 2>&1; tail -3 $o/profile_$tag.log
python3 scripts/pmc_summary.py $o/prof_$tag > $o/rocprof_$tag.md 2>/dev/null; head -20 $o/rocprof_$tag.md
