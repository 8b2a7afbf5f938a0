#!/bin/bash
# batch / nsplit sweep of the whole pipeline (frames per step, sub-batches); optional env prefix per config
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
DEF="6144:4 5120:4 4096:4 7168:4 8192:4 6144:3 9216:6"
for cfg in ${SWEEP:-$DEF}; do
  b=${cfg%%:*}; n=${cfg##*:}
  r=$(timeout 600 python bench.py --no-cpu-baseline --no-extras --batch $b --nsplit $n --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "batch $b nsplit $n: $r"
done | tee gpurun_out/sweep_final.log
