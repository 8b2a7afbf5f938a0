#!/bin/bash
# headline workload over (batch, sub-batches): how many frames should be resident, and in how many launches?  gpurun_out/batch_sweep.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
show='import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernel_ms_per_launch"]["k_lsd_grow"])'
for cfg in "6144 4" "7168 4" "6144 3" "6144 6" "8192 4" "5120 4" "7168 7" "6144 4"; do
set -- $cfg
echo -n "batch $1 nsplit $2: " | tee -a gpurun_out/batch_sweep.txt
timeout 300 python bench.py --steps 8 --warmup 2 --batch $1 --nsplit $2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a gpurun_out/batch_sweep.txt
done
