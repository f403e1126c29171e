#!/bin/bash
# builds tools/conv_bench (tuning harness) next to the library objects
set -e
cd "$(dirname "$0")/.."
P=coupe/optical_flow_based_deep_video_stabilization_amd
python $P/build.py >/dev/null
for abl in 0 1 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVSTAB_HARNESS -DVSTAB_ABL=$abl tools/conv_bench.hip $P/csrc/conv_mfma.hip $P/csrc/conv_rowwin.hip $P/csrc/pack.cpp $P/csrc/api.cpp $P/csrc/flow_ops.hip $P/csrc/sampler_ops.hip $P/csrc/nldf_ops.hip $P/csrc/clip_ops.hip $P/csrc/nldf_api.cpp -o tools/conv_bench_abl$abl 2>/tmp/bt_$abl.log &
done
wait
echo built tools/conv_bench_abl*
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/warp_bench.hip -o tools/warp_bench
echo built tools/warp_bench
