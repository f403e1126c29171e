#!/bin/bash
# round 3, visit b: full GPU suite, then the self-launch study: plain run, four sequential self-launched
# single-rank RCCL runs, and ONE run with the old OMP_NUM_THREADS = os.cpu_count() to confirm the cause of the host-bound runs
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
python3 - <<'PY' > gpurun_out/host_facts_r3b.txt 2>&1
import os
print("os.cpu_count()", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "-", e.__class__.__name__)
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
import torch
print("torch.get_num_threads()", torch.get_num_threads())
PY
cat gpurun_out/host_facts_r3b.txt
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/pytest_r3b.log 2>&1; rc=$?
tail -n 15 gpurun_out/pytest_r3b.log; echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
B="--no-cpu-baseline --steps 40 --warmup 10"
timeout -k 10 300 python3 bench.py $B > gpurun_out/sl_plain.json 2> gpurun_out/sl_plain.err || { tail gpurun_out/sl_plain.err; exit 1; }
python3 -c "import json;d=json.load(open('gpurun_out/sl_plain.json'));print('plain', d['ms_per_step'], d['value'])"
timeout -k 10 300 python3 bench.py $B --alloc-per-step > gpurun_out/sl_plain_alloc.json 2> gpurun_out/sl_plain_alloc.err || exit 1
python3 -c "import json;d=json.load(open('gpurun_out/sl_plain_alloc.json'));print('plain alloc-per-step', d['ms_per_step'], d['value'])"
export VSTAB_FORCE_DIST=1 VSTAB_BENCH_DEBUG=1
for i in 1 2 3 4; do
  timeout -k 10 300 python3 bench.py --gpus 1 $B > gpurun_out/sl_dist_$i.json 2> gpurun_out/sl_dist_$i.err; r=$?
  echo "self-launched run $i rc=$r"; [ $r -eq 0 ] || { tail -20 gpurun_out/sl_dist_$i.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/sl_dist_$i.json'));print('  dist', d['ms_per_step'], d['value'])"; grep "timed region\|self-launch" gpurun_out/sl_dist_$i.err | cut -c1-260
done
export OMP_NUM_THREADS=$(python3 -c "import os;print(os.cpu_count())")
timeout -k 10 300 python3 bench.py --gpus 1 $B > gpurun_out/sl_dist_oldomp.json 2> gpurun_out/sl_dist_oldomp.err; echo "old-OMP run rc=$?"
python3 -c "import json;d=json.load(open('gpurun_out/sl_dist_oldomp.json'));print('  dist old OMP=$OMP_NUM_THREADS', d['ms_per_step'], d['value'])"; grep "timed region" gpurun_out/sl_dist_oldomp.err
timeout -k 10 300 python3 bench.py --gpus 1 $B --alloc-per-step > gpurun_out/sl_dist_oldomp_alloc.json 2> gpurun_out/sl_dist_oldomp_alloc.err; echo "old-OMP alloc-per-step run rc=$?"
python3 -c "import json;d=json.load(open('gpurun_out/sl_dist_oldomp_alloc.json'));print('  dist old OMP alloc', d['ms_per_step'], d['value'])"; grep "timed region" gpurun_out/sl_dist_oldomp_alloc.err
exit 0
