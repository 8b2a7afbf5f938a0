#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job8
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/dbg/paranoid_check.py libplslam_hip_profpar2.so 1024 2>&1 | grep -v amdgpu.ids | tee $O/paranoid2_counter.txt
timeout 1500 python tools/dbg/paranoid_check.py libplslam_hip_par.so 1024 2>&1 | grep -v amdgpu.ids | tee $O/paranoid_product.txt
timeout 1500 python tools/dbg/paranoid_check.py libplslam_hip_profpar.so 1024 2>&1 | grep -v amdgpu.ids | tee $O/paranoid_counter.txt
exit 0
