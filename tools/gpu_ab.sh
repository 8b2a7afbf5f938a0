#!/bin/bash
# A/B of two builds of the library on the line extractor (k_lsd_grow dominated): PLSLAM_HIP_LIB selects the build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
export PLH_GROW_MW_WAVES=0
for rep in 1 2 3; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
for b in 1536 6144; do
echo -n "rep $rep $lib batch $b: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python tools/grow_prof.py --batch $b --reps 4 2>&1 | grep "line extract" | tee -a $O/ab.txt
done
done
done
exit 0
