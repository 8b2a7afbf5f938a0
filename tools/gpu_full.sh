#!/bin/bash
# the round-end sequence the driver runs: GPU parity suite, smoke, default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/full_smoke.txt
timeout 1200 python bench.py 2>gpurun_out/full_bench.err | tail -1 > gpurun_out/full_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/full_bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
print('latency', d.get('latency_ms_single_frame'))
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
print('streaming', d.get('streaming',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
exit 0
