#!/bin/bash
# Build a variant of the product library for an A/B job: tools/build_variant.sh NAME [-DFLAG=..]...  ->  pl-slam_amd/libplslam_hip_NAME.so
# (same flags as __graft_entry__.build_hip; the variant carries the build id of the sources it was made from + ":NAME")
set -eu
cd "$(dirname "$0")/.."
name=$1; shift
sid=$(python -c "import __graft_entry__ as g; print(g.source_id())")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-function \
  "$@" "-DPLH_BUILD_ID=\"$sid\"" -o pl-slam_amd/libplslam_hip_$name.so pl-slam_amd/csrc/*.hip
echo "built pl-slam_amd/libplslam_hip_$name.so ($*)"
