#!/bin/bash
# round 4: two-octave LINEextractor on the GPU, goldens of the reference's line path, device sincos against the host, A/B of
# k_lsd_grow built for 7 and for 6 wavefronts per SIMD.  gpurun_out/r4e/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4e
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py tests/test_ref_line.py tests/test_frontend_example.py tests/test_abi.py -m gpu -x -q --timeout 800 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.txt
g++ -O2 -march=x86-64-v3 -ffp-contract=off -std=c++17 -c -o /tmp/sincos_cr_host.o tools/ubench/sincos_cr_host.cc && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -c -o /tmp/sincos_cr_dev.o tools/ubench/sincos_cr_dev.hip && \
/opt/rocm/bin/hipcc -o /tmp/sincos_cr_dev /tmp/sincos_cr_dev.o /tmp/sincos_cr_host.o -lpthread && timeout 600 /tmp/sincos_cr_dev | tee $O/sincos_cr_dev.txt
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "rep $rep $lib headline: " | tee -a $O/ab_waves.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/ab_waves.txt
done
done
exit 0
