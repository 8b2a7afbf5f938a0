#!/bin/bash
# quick check: line + e2e GPU tests, bench, SQ counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "line or e2e or adaptor" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/bench_quick.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print(d['value'], 'frames/s;', 'roofline', d['roofline']['frac'], 'fast', d['roofline_fast']['frac'])
PY
timeout 900 bash tools/pmc_insts.sh 256 2>&1 | tail -24 | head -6
exit 0
