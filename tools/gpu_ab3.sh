#!/bin/bash
# three builds on the configs[4] share and on the 512-frame line extractor alone
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab5.so libplslam_hip_ab6.so; do
echo -n "rep $rep $lib share512: " | tee -a $O/ab3.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'].get('k_lsd_grow'))" | tee -a $O/ab3.txt
done
done
exit 0
