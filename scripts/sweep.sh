cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
VSTAB_TAB_SPLIT=1 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/$/ (split)/'
done
