#!/bin/bash
# A/B of two builds of the library on multi-wavefront region growing: one frame (latency) and 512 frames, 8 waves per frame
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
export PLH_GROW_MW_WAVES=${1:-8}
for rep in 1 2 3; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
for b in 1 512; do
echo -n "rep $rep $lib batch $b: " | tee -a $O/ab_mw.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python tools/grow_prof.py --batch $b --reps 6 2>&1 | grep "line extract" | tee -a $O/ab_mw.txt
done
done
done
exit 0
