#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest orb+line gpu"
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_ref_orb.py tests/test_line.py tests/test_ref_line.py -m gpu -x -q --timeout 600 2>&1 | tail -4 | tee gpurun_out/pytest_orb.log
echo "== orb kernel times (1024 frames): pyramid, FAST, octree, orient+brief"
timeout 300 python tools/orb_time.py 2>&1 | tail -1 | tee gpurun_out/orb_time4.log
echo "== line bench 1024"
timeout 300 python tools/line_bench.py 1024 2>&1 | tail -1 | tee gpurun_out/line_bench.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step'); print(d['kernel_ms_per_launch'])" | tee gpurun_out/bench_short.log
exit 0
