#!/bin/bash
# stream layout experiments for the pipelined front end (PLSLAM_PIPE, pl-slam_amd/pipeline.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for m in default eqprio eqprio0 same shared_orb; do
  r=$(PLSLAM_PIPE=$m timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "mode $m: $r"
done | tee gpurun_out/pipe_modes.log
for m in default same; do for ns in 6 8; do
  r=$(PLSLAM_PIPE=$m timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 10 --nsplit $ns 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "mode $m nsplit $ns: $r"
done; done | tee -a gpurun_out/pipe_modes.log
