#!/bin/bash
# Round 6, job 7: the protocol's claim checked where it is made -- -DPLH_MW_PARANOID builds (product source; counter-build source).
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job7
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python tools/dbg/paranoid_check.py libplslam_hip_par.so 1024 2>&1 | grep -v amdgpu.ids | tee $O/paranoid_product.txt
timeout 1200 python tools/dbg/paranoid_check.py libplslam_hip_profpar.so 1024 2>&1 | grep -v amdgpu.ids | tee $O/paranoid_counter.txt
exit 0
