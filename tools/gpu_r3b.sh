#!/bin/bash
# round 3, throughput kernel: sincos accuracy evidence, line / soak / adaptor parity, phase profile, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/r3b
mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pl-slam_amd/csrc -I include -o /tmp/sincos_ulp tools/ubench/sincos_ulp.hip 2>/dev/null && /tmp/sincos_ulp | tee $O/sincos_ulp.txt
timeout 1500 python -m pytest tests/test_line.py tests/test_soak_gpu.py tests/test_adaptor_exec.py -m gpu -x -q --timeout 900 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $O/tests.txt
PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_prof2.so PLH_GROW_MW_WAVES=0 timeout 600 python tools/grow_prof.py --batch 256 2>&1 | tail -20 | tee $O/grow_prof_256.txt
PLH_GROW_MW_WAVES=0 timeout 600 python tools/grow_prof.py --batch 1536 2>&1 | head -3 | tee $O/line_1536.txt
PLH_GROW_MW_WAVES=0 timeout 600 python tools/grow_prof.py --batch 6144 2>&1 | head -3 | tee $O/line_6144.txt
timeout 1200 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], 'verified', d.get('verified'))
PY
exit 0
