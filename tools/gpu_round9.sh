#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== batch / nsplit sweep (1536 frames per sub-batch)"
for cfg in "9216 6" "12288 8" "12288 4" "18432 12"; do
  set -- $cfg
  timeout 900 python bench.py --steps 5 --warmup 2 --batch $1 --nsplit $2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 nsplit $2 ->', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', d['kernel_ms_per_launch_timed_region']['k_lsd_grow'])" | tee -a gpurun_out/sweep2.log
done
exit 0
