#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/mw4.txt
for gap in ${GAPS:-1 4 8 16 32}; do
  echo "gap $gap" | tee -a gpurun_out/mw4.txt
  PLH_GROW_MW_GAP=$gap timeout 600 python tools/mw_sweep.py --batches ${BATCHES:-1,512} --waves ${WAVES:-8} --reps 5 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/mw4.txt
done
exit 0
