#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "line or e2e or adaptor" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/bench_r15.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r15.json'))
print(d['value'], 'frames/s;', d['ms_per_step'], 'ms/step; roofline', d['roofline']['frac'], 'fast', d['roofline_fast']['frac'])
PY
timeout 300 python tools/halves.py 6144 4 2>&1 | tail -3
timeout 900 bash tools/pmc_insts.sh 256 2>&1 | tail -24 | head -12
timeout 1200 bash tools/pmc_traffic.sh 256 > gpurun_out/pmc_traffic.log 2>&1; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/pmc/traffic.json'))
    for k in ('k_lsd_grow','k_lsd_grad','k_lsd_bin_hist','k_lsd_bin_scatter'):
        print(k, d['kernels'].get(k))
except Exception as e: print('traffic failed', e)
PY
exit 0
