#!/bin/bash
# A/B of two builds of the library in one job (pl-slam_amd/libplslam_hip.so = the tree's, libplslam_hip_ab.so = the other one):
# line tests of the tree's build first, then per build the headline, the region-growing kernel at full residency in one launch
# (--serial --nsplit 1) and the configs[4] share.  gpurun_out/ablib/ab.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/ablib
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
# QUICK=1: no tests, no configs[4] share (for changes that cannot alter results)
[ "${QUICK:-0}" = 1 ] || timeout 900 python -m pytest tests/test_line.py tests/test_frontend_example.py -m gpu -x -q --timeout 800 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; print(d["value"], d["ms_per_step"], {n: k[n] for n in k if "grow" in n or "prep" in n or "rects" in n})'
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "rep $rep $lib headline: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
echo -n "rep $rep $lib 6144 in one launch, serial: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 3 --warmup 1 --batch 6144 --nsplit 1 --serial --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
[ "${QUICK:-0}" = 1 ] && exit 0
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "$lib share512: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
exit 0
