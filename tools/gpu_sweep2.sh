#!/bin/bash
# batch / nsplit sweep of the whole pipeline (frames per step, sub-batches)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "6144 4" "5120 4" "4096 4" "6144 6" "6144 3" "7680 5" "5120 5" "6144 8"; do
  set -- $cfg
  r=$(timeout 600 python bench.py --no-cpu-baseline --no-extras --batch $1 --nsplit $2 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "batch $1 nsplit $2: $r"
done | tee gpurun_out/sweep_r13.log
