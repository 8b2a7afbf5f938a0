#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest adaptor exec + abi (gpu)"
timeout 900 python -m pytest tests/test_adaptor_exec.py tests/test_abi.py tests/test_frame_search.py -m gpu -x -q --timeout 600 2>&1 | tail -8 | tee gpurun_out/pytest_adaptor.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench.log
echo "== instruction budget per kernel"
timeout 900 bash tools/pmc_insts.sh 256 2>&1 | tail -30 | tee gpurun_out/pmc_insts.log
exit 0
