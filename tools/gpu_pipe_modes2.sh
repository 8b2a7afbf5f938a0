#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "same 6144 4" "same 6144 3" "same 4608 3" "same 6144 2" "same 7680 5" "same 8192 4" "same 9216 6" "eqprio 6144 4" "eqprio 6144 3"; do
  set -- $cfg
  r=$(PLSLAM_PIPE=$1 timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 --batch $2 --nsplit $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "mode $1 batch $2 nsplit $3: $r"
done | tee gpurun_out/pipe_modes2.log
