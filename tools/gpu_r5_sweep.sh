#!/bin/bash
# Round 5: the ADV headline over (resident batch, sub-batches).  gpurun_out/r5sweep/sweep.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5sweep
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for cfg in "6144 4" "6144 3" "6144 6" "6144 8" "7168 4" "7168 7" "8192 4" "4096 4"; do
  set -- $cfg
  printf "batch %5d nsplit %d : " $1 $2 | tee -a $O/sweep.txt
  timeout 600 python bench.py --batch $1 --nsplit $2 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f f/s %.2f ms  box %s" % (d["value"], d["ms_per_step"], (d.get("box") or {}).get("probe_ms")))' | tee -a $O/sweep.txt
done
exit 0
