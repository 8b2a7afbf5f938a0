#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "line or e2e or adaptor" 2>&1 | tail -3
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'frames/s;', d['ms_per_step'], 'ms/step; roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('ms_per_launch_alone'))"; done
timeout 300 python tools/halves.py 6144 4 2>&1 | tail -3
bash tools/pmc_grow_mem.sh 2>&1 | tail -2
