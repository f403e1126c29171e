#!/bin/bash
# diagnostic build of the library with in-kernel stamps (scripts/insitu_stamps.py): NOT the product library
set -e
cd "$(dirname "$0")/.."
P=coupe/optical_flow_based_deep_video_stabilization_amd
SRCS=$(python -c "from coupe.optical_flow_based_deep_video_stabilization_amd import build; print(' '.join('$P/csrc/' + s for s in build.SOURCES))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -shared -DVSTAB_HARNESS -DVSTAB_STAMP $SRCS -o tools/libvstab_hip_stamp.so
echo built tools/libvstab_hip_stamp.so
