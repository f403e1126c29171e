#!/bin/bash
# bench lines for the other BASELINE configs (not the headline): usage scripts/gpu_configs.sh <tag>
tag=${1:-run}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python3 bench.py --batch 1 --height 256 --width 256 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/cfg0_$tag.json 2> gpurun_out/cfg0_$tag.err && tail -2 gpurun_out/cfg0_$tag.err
python3 bench.py --batch 32 --height 720 --width 1280 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/cfg2_$tag.json 2> gpurun_out/cfg2_$tag.err && tail -2 gpurun_out/cfg2_$tag.err
python3 bench.py --batch 16 --height 1080 --width 1920 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/cfg4_$tag.json 2> gpurun_out/cfg4_$tag.err && tail -2 gpurun_out/cfg4_$tag.err
python3 bench.py --batch 8 --height 512 --width 512 --cin 6 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/cfg1c6_$tag.json 2> gpurun_out/cfg1c6_$tag.err && tail -2 gpurun_out/cfg1c6_$tag.err
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/cfg*_$tag.json
