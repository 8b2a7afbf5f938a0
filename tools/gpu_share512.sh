#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/full_tests.txt
for ns in 1 2 4; do
timeout 600 python bench.py --batch 512 --nsplit $ns --rows 376 --cols 1241 --nfeatures 2000 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('nsplit $ns', d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])"
done
exit 0
