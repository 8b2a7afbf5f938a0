#!/bin/bash
# resident batch x sub-batches of the headline workload on the tree's build: gpurun_out/sweep/sweep.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/sweep
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for cfg in ${CFGS:-6144:4 7168:4 7680:5 6144:3 8192:4 9216:6}; do
set -- ${cfg/:/ }
echo -n "rep $rep batch $1 nsplit $2: " | tee -a $O/sweep.txt
timeout 600 python bench.py --batch $1 --nsplit $2 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/sweep.txt
done
done
exit 0
