#!/bin/bash
# tile x split-K sweep of the split-K layers at the headline batch (tools/conv_bench_abl0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export VSTAB_BENCH_FILL=zero
{
for l in 4 6 8 10 11 12; do
  for t in 0 1 3 4; do
    for ks in 1 2 3 4 5 6 8 10 12 16; do
      timeout -k 5 30 tools/conv_bench_abl0 $l 8 512 512 $t $ks 20 2>&1 | grep -E "^layer|error|HIP"
    done
  done
  echo "progress: layer $l done" >&2
done
} > gpurun_out/sweep_b8_512x512.log
