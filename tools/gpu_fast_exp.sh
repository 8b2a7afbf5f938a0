#!/bin/bash
# FAST kernel experiment round: ORB parity tests, per-kernel times for several strip widths, SQ instruction counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest orb gpu"
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_ref_orb.py tests/test_e2e_gpu.py -m gpu -x -q --timeout 600 2>&1 | tail -8 | tee gpurun_out/pytest_orb.log
for w in 100 140 200 270 330 660; do
  echo "== strip width $w"
  PLH_FAST_STRIP_W=$w timeout 300 python tools/orb_time.py 2>&1 | tail -1 | tee -a gpurun_out/orb_time.log
done
echo "== pmc orb"
PLH_FAST_STRIP_W=200 timeout 600 bash tools/pmc_orb.sh 2>&1 | tail -8 | tee gpurun_out/pmc_orb.log
exit 0
