#!/bin/bash
# scratch: slim address arithmetic
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for f in zero rand; do
export VSTAB_BENCH_FILL=$f
echo "== fill $f"
for l in 1 2 3 5; do timeout -k 10 60 tools/conv_bench_abl0 $l 8 512 512 0 1 30 || exit 1; done
timeout -k 10 60 tools/conv_bench_abl0 13 8 512 512 1 1 30 || exit 1
done
} > gpurun_out/sweep.log 2>&1
