#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py -m gpu -x -q --timeout 300 2>&1 | tail -5 | tee gpurun_out/mw2_tests.txt
timeout 600 python tools/mw_sweep.py --batches ${BATCHES:-1,64,512} --waves ${WAVES:-0,4,8,16} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mw2_sweep.txt | tail -30
rm -f gpurun_out/mwprof.txt
for b in 1 512; do
PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof.so timeout 600 python tools/mw_prof.py --batch $b --waves ${WAVES:-0,4,8,16} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/mwprof.txt
done
exit 0
