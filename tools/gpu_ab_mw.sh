#!/bin/bash
# A/B of two builds on what the multi-wavefront region-growing kernel serves: line + soak tests of the tree's build, then per build
# one-frame latency (tools/latency.py), the line extractor at 64 / 512 frames (tools/mw_sweep.py) and the configs[4] share.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/abmw
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
[ "${QUICK:-0}" = 1 ] || timeout 900 python -m pytest tests/test_line.py tests/test_soak_gpu.py tests/test_frontend_example.py -m gpu -x -q --timeout 800 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; print(d["value"], d["ms_per_step"], {n: k[n] for n in k if "grow" in n})'
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo "rep $rep $lib" | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 300 python tools/latency.py 2>/dev/null | grep -v amdgpu | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 300 python tools/mw_sweep.py --batches 64,512 --waves 0,8 --reps 3 2>/dev/null | grep batch | tee -a $O/ab.txt
echo -n "share512: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
exit 0
