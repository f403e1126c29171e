cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_vgg.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; echo "(asm reads)"
done
python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "conv|deconv" | cut -c1-100
