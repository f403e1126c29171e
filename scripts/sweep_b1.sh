#!/bin/bash
# tile x split-K sweep of every conv-like layer at one sample (tools/conv_bench_abl0): usage sweep_b1.sh B H W
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B=${1:-1}; H=${2:-384}; W=${3:-512}
export VSTAB_BENCH_FILL=zero
{
for l in 1 2 3 4 5 6 7 8 9 10 11 12 13; do
  for t in 0 1 3 4; do
    for ks in 1 2 3 4 5 6 8 10 12 16 20 24 32 40 64; do
      timeout -k 5 30 tools/conv_bench_abl0 $l $B $H $W $t $ks 20 2>&1 | grep -E "^layer|error|HIP" 
    done
  done
  echo "progress: layer $l done" >&2
done
} > gpurun_out/sweep_b${B}_${H}x${W}.log
