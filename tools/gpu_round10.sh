#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in libplslam_hip.so libplslam_hip_w7.so libplslam_hip_w8.so; do
  echo "== $lib"
  PLSLAM_HIP_LIB=$PWD/pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step', 'grow alone', d['kernel_ms_per_launch']['k_lsd_grow'], 'pyr', d['kernel_ms_per_launch']['k_pyr_down x7'])" | tee -a gpurun_out/waves_exp.log
  PLSLAM_HIP_LIB=$PWD/pl-slam_amd/$lib timeout 600 python bench.py --steps 6 --warmup 2 --batch 8192 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  batch 8192:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" | tee -a gpurun_out/waves_exp.log
done
exit 0
