#!/bin/bash
# first GPU contact of k_lsd_grow_mw: parity tests, then the wavefront / batch sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py -m gpu -x -q --timeout 300 2>&1 | tail -5 | tee gpurun_out/mw1_tests.txt
timeout 600 python tools/mw_sweep.py --batches 1,8,64,512 --waves 0,2,4,8,16 2>&1 | tee gpurun_out/mw1_sweep.txt | tail -30
exit 0
