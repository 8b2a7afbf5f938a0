#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/mw3.txt
for lag in ${LAGS:-4 8 16 32 64}; do
  echo "lag $lag" | tee -a gpurun_out/mw3.txt
  PLH_GROW_MW_LAG=$lag timeout 600 python tools/mw_sweep.py --batches ${BATCHES:-1,512} --waves ${WAVES:-4,8} --reps 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/mw3.txt
done
exit 0
