#!/bin/bash
# round-2 visit F: sustained headline run, training line, HBM-side profile at B=16 1080p and at cfg1 (stats + PMC passes)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline > gpurun_out/r2f_sustained.json 2> gpurun_out/r2f_sustained.err; rc=$?
tail -n 3 gpurun_out/r2f_sustained.err; cut -c1-260 gpurun_out/r2f_sustained.json; echo "sustained rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench_train.py --steps 10 --warmup 3 > gpurun_out/r2f_train.json 2> gpurun_out/r2f_train.err; rc=$?
cut -c1-300 gpurun_out/r2f_train.json; echo "train rc=$rc"
[ $rc -ne 0 ] && exit $rc
for cfg in "1080 --batch 16 --height 1080 --width 1920 --steps 4 --warmup 1" "cfg1 --steps 10 --warmup 3"; do
  set -- $cfg; tag=$1; shift
  out=gpurun_out/r2f_prof_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/stats.log || { echo "stats pass $tag failed"; tail -5 $out/stats.log; exit 1; }
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
    timeout -k 10 600 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/pmc_$name -- python3 bench.py "$@" --no-cpu-baseline --no-kernel-events > $out/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $out/pmc_$name.log; exit 1; }
  done
  find $out -name '*.csv' -size +8M -delete
  echo "profile $tag done"
done
