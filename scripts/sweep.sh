#!/bin/bash
# scratch: tile experiments on the GPU box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for l in 13 1 12; do
  for t in 0 1 3; do
    timeout -k 10 60 tools/conv_bench_abl0 $l 8 512 512 $t -1 30 || exit 1
  done
done
} > gpurun_out/sweep.log 2>&1
