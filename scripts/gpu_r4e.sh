#!/bin/bash
# round 4, visit e: full -m gpu suite, the N>1 rehearsal with as many ranks as a one-GPU box admits (6 processes may use the card: the
# 8-rank shapes run under gloo on the CPU, tests/test_distributed_cpu.py), training line with its roofline, the homography warp's trace
# (no library GEMM in front of it any more), A/B of the weight-stream kernel's split-K cap.
set -u
tag=${1:-r04e}
mkdir -p gpurun_out/r04_world; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
if [ -z "${SKIP_TESTS:-}" ]; then timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/pytest_$tag.log 2>&1; rc=$?; else rc=0; fi
tail -n 12 gpurun_out/pytest_$tag.log | cut -c1-300; echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
# four self-launched gloo ranks sharing the GPU (six ranks + their launcher were seven processes with the GPU open: the box admits six): numbers mean nothing, rc 0 and ONE JSON line are the point
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --steps 7 --warmup 3 --batch 1 --height 128 --width 128 --gather-every 4 --no-cpu-baseline > gpurun_out/r04_world/bench_gloo4.json 2> gpurun_out/r04_world/bench_gloo4.err; echo "bench gloo4 rc=$? lines=$(grep -c '^{' gpurun_out/r04_world/bench_gloo4.json)"
timeout -k 10 400 python3 bench.py --gpus 4 --backend gloo --steps 7 --warmup 3 --batch 1 --height 128 --width 128 --gather-every 4 --gather-schedule direct --no-cpu-baseline > gpurun_out/r04_world/bench_gloo4_direct.json 2> gpurun_out/r04_world/bench_gloo4_direct.err; echo "bench gloo4 direct rc=$? lines=$(grep -c '^{' gpurun_out/r04_world/bench_gloo4_direct.json)"
timeout -k 10 400 python3 bench_clip.py --gpus 4 --backend gloo --frames 67 --height 128 --width 128 --micro-batch 4 --check > gpurun_out/r04_world/bench_clip_gloo4.json 2> gpurun_out/r04_world/bench_clip_gloo4.err; echo "bench_clip gloo4 rc=$? $(cut -c1-200 gpurun_out/r04_world/bench_clip_gloo4.json)"
# training line
timeout -k 10 400 python3 bench_train.py --steps 10 --warmup 3 --phases > gpurun_out/bench_train_$tag.json 2> gpurun_out/bench_train_$tag.err || tail -5 gpurun_out/bench_train_$tag.err
cut -c1-1500 gpurun_out/bench_train_$tag.json
# homography warp under the profiler: which kernels run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_st_$tag -- python3 scripts/st_bench.py --shapes 32x720x1280 --kinds stab --iters 20 > gpurun_out/prof_st_$tag.log 2>&1 || tail -5 gpurun_out/prof_st_$tag.log
find gpurun_out/prof_st_$tag -name '*kernel_stats.csv' | head -1 | xargs -r head -8 | cut -c1-160
find gpurun_out/prof_st_$tag -name '*.csv' -size +4M -delete
# HISTORICAL: this visit measured a build with an experiment flag (4) that is gone -- vstab_set_plan_flags now rejects every bit outside 1|2;
# the line below is the same A/B with the flags that still exist
bash scripts/gpu_ab_flags.sh $tag "0 1" 2 | grep "^round"
