#!/bin/bash
# an A/B build of the product library with extra -D flags: usage build_variant_lib.sh <name> [-DFLAG ...]  ->  tools/libvstab_hip_<name>.so
# (select it with VSTAB_LIB=tools/libvstab_hip_<name>.so; scripts/ab_bench.sh interleaves two environments in one box visit)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
P=coupe/optical_flow_based_deep_video_stabilization_amd
SRCS=$(python -c "from coupe.optical_flow_based_deep_video_stabilization_amd import build; print(' '.join('$P/csrc/' + s for s in build.SOURCES))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -shared "$@" $SRCS -o tools/libvstab_hip_$NAME.so
echo built tools/libvstab_hip_$NAME.so
