#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_soak_gpu.py -m gpu -x -q -s --timeout 900 2>&1 | grep -v "amdgpu.ids" | grep -E "soak|waves|passed|failed|Error|error|assert" | tee gpurun_out/soak.txt
true
exit 0
