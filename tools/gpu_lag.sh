#!/bin/bash
# run-ahead limit and commit batching of k_lsd_grow_mw on the configs[4] share (512 frames of 1241x376)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/lag
mkdir -p $O
export TMPDIR=/tmp
for lag in 128 256 448; do
for gap in 4 8 16; do
PLH_GROW_MW_LAG=$lag PLH_GROW_MW_GAP=$gap timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lag $lag gap $gap', d['value'], d['ms_per_step'], d['kernel_ms_per_launch'].get('k_lsd_grow'))" | tee -a $O/lag.txt
done
done
exit 0
