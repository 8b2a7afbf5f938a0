#!/bin/bash
# how many frames resident, in how many sub-batches: full-residency region-growing launches (6144 frames each) against the default 4 x 1536
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/resid
mkdir -p $O
export TMPDIR=/tmp
for cfg in "6144 4" "6144 1" "12288 2" "12288 4" "18432 3" "24576 4"; do
set -- $cfg
timeout 900 python bench.py --batch $1 --nsplit $2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>$O/err_$1_$2.txt | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('batch $1 nsplit $2', d['value'], d['ms_per_step'])
except Exception as e: print('batch $1 nsplit $2 failed', e)" | tee -a $O/resid.txt
done
exit 0
