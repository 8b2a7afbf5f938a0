#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v "^  \|^$" | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== bench (default flags)"
( time timeout 1200 python bench.py ) 2>&1 | tail -6 | tee gpurun_out/bench.log
echo "== bench --force-dist (1-rank RCCL rehearsal of the N > 1 path)"
timeout 600 python bench.py --steps 4 --warmup 1 --force-dist --no-extras --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1500 | tee gpurun_out/fd.log
exit 0
