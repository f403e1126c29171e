#!/bin/bash
# scratch: 64x128 tile for tiny-M layers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== B=1 384x512 tile0 vs tile3"
for l in 6 7; do for ks in 12 16 21 24 32 42; do for t in 0 3; do timeout -k 10 60 tools/conv_bench_abl0 $l 1 384 512 $t $ks 50 || exit 1; done; done; done
for l in 8 9 10; do for ks in 8 12 16 24 32; do for t in 0 3; do timeout -k 10 60 tools/conv_bench_abl0 $l 1 384 512 $t $ks 50 || exit 1; done; done; done
for l in 4 5 11; do for ks in 4 6 9 12; do for t in 0 3; do timeout -k 10 60 tools/conv_bench_abl0 $l 1 384 512 $t $ks 50 || exit 1; done; done; done
echo "== B=8 512x512"
for l in 8 9; do for ks in 4 8 16; do for t in 0 3; do timeout -k 10 60 tools/conv_bench_abl0 $l 8 512 512 $t $ks 30 || exit 1; done; done; done
} > gpurun_out/sweep.log 2>&1
