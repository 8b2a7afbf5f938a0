#!/bin/bash
# Round 6, job 5: is the counter build's disagreement on 10 of the soak's 1024 frames the same from run to run?  + the GPU suite on this build.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job5
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/dbg/prof_mismatch.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/prof_mismatch.txt
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.txt
exit 0
