cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vggprof -- python3 bench.py --batch 2 --height 1080 --width 1920 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --vgg16 > gpurun_out/vggprof.log 2>&1
find gpurun_out/vggprof -name '*kernel_stats.csv' | xargs head -12
