#!/bin/bash
# configs[4] share (512 frames of 1241x376 per GPU): sub-batch count x waves per frame
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c
mkdir -p $O
export TMPDIR=/tmp
for w in 8 16 4; do
for ns in 1 2 4; do
PLH_GROW_MW_WAVES=$w timeout 600 python bench.py --batch 512 --nsplit $ns --rows 376 --cols 1241 --nfeatures 2000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('waves $w nsplit $ns', d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/share512_sweep.txt
done
done
exit 0
