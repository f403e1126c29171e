cd $GRAFT_REPO_ROOT
(python3 bench.py --steps 600 --warmup 5 --no-cpu-baseline > gpurun_out/long.json 2> gpurun_out/long.err) &
BP=$!
sleep 14
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8
  echo ---
  sleep 0.4
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' gpurun_out/long.json | head -1
