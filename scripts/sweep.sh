cd $GRAFT_REPO_ROOT
for i in 1 2; do
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>gpurun_out/ev.err | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*' | head -3 | tr '\n' ' '; echo "(events)"
done
tail -18 gpurun_out/ev.err
