#!/bin/bash
# builds the tuning harnesses under tools/ (not part of the product): conv_bench with the VSTAB_HARNESS switches and ablations, warp_bench
set -e
cd "$(dirname "$0")/.."
P=coupe/optical_flow_based_deep_video_stabilization_amd
python $P/build.py >/dev/null
SRCS=$(python -c "from coupe.optical_flow_based_deep_video_stabilization_amd import build; print(' '.join('$P/csrc/' + s for s in build.SOURCES))")
# -DVSTAB_STAMP (one more build, tools/conv_bench_stamp): s_memtime stamps of every workgroup at entry / loop start / loop end / exit
for abl in ${ABLS:-0 1 5}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVSTAB_HARNESS -DVSTAB_ABL=$abl tools/conv_bench.hip $SRCS -o tools/conv_bench_abl$abl 2>/tmp/bt_$abl.log &
done
wait
echo built tools/conv_bench_abl*
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/warp_bench.hip -o tools/warp_bench
echo built tools/warp_bench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVSTAB_HARNESS -DVSTAB_ABL=0 -DVSTAB_STAMP tools/conv_bench.hip $SRCS -o tools/conv_bench_stamp
echo built tools/conv_bench_stamp
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shape.hip -o tools/mfma_shape 2>/dev/null
echo built tools/mfma_shape
