#!/bin/bash
# A/B of two builds of the library on the whole front end: the headline workload and the configs[4] share
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "rep $rep $lib headline: " | tee -a $O/ab_bench.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/ab_bench.txt
echo -n "rep $rep $lib share512: " | tee -a $O/ab_bench.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/ab_bench.txt
done
done
exit 0
