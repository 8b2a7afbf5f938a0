#!/bin/bash
# round-2 instruction-cut check: parity suite, bench, per-kernel SQ instruction counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v "^  \|^$" | tail -12 | tee gpurun_out/pytest_gpu.log
echo "== bench"
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/bench_r11.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r11.json'))
print(d['value'], 'frames/s;', 'roofline', d['roofline']['frac'], 'fast', d['roofline_fast']['frac'])
PY
echo "== SQ instruction counters"
timeout 900 bash tools/pmc_insts.sh 256 2>&1 | tail -24 | tee gpurun_out/pmc_insts.log
exit 0
