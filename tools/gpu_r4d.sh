#!/bin/bash
# round 4: line parity on the W-cached rect kernel + correctly rounded sincos, device sincos against the host, A/B of k_lsd_grow built
# for 8 and for 7 wavefronts per SIMD.  gpurun_out/r4d/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4d
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py tests/test_soak_gpu.py -m gpu -x -q -s --timeout 800 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|soak" | tee $O/tests.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o /tmp/sincos_cr_dev tools/ubench/sincos_cr_dev.hip && timeout 600 /tmp/sincos_cr_dev | tee $O/sincos_cr_dev.txt
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "rep $rep $lib headline: " | tee -a $O/ab_waves.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/ab_waves.txt
done
done
exit 0
