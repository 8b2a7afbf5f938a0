#!/bin/bash
# Round 6, job 6: is the multi-wavefront protocol timing-sensitive?  The PRODUCT build with pseudo-random pauses at its hand-over points
# (-DPLH_MW_JITTER) against the product build, 1024 x 8 and 4 x (256 x 8), three runs; then where a frame's wavefronts spend their time.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job6
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/dbg/prof_mismatch.py 1024 libplslam_hip_jit.so 2>&1 | grep -v amdgpu.ids | tee $O/jitter_mismatch.txt
PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so timeout 600 python tools/mw_prof.py --batch 1 --waves 8,16 2>&1 | grep -v amdgpu.ids | tee $O/mw_prof_1.txt
PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so timeout 600 python tools/mw_prof.py --batch 512 --waves 8 2>&1 | grep -v amdgpu.ids | tee $O/mw_prof_512.txt
exit 0
