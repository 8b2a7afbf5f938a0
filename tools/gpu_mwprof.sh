#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/mwprof.txt
for b in ${BATCHES:-1 512}; do
PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so timeout 600 python tools/mw_prof.py --batch $b --waves ${WAVES:-0,4,8} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/mwprof.txt
done
exit 0
