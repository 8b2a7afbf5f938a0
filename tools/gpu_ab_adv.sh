#!/bin/bash
# A/B of two builds on LSD_REFINE_ADV: the ADV tests and soak of the tree's build, then per build the headline workload with
# --refine adv and the rocprofv3 statistics of the ADV kernels.  gpurun_out/abadv/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/abadv
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_line.py tests/test_soak_gpu.py tests/test_frontend_example.py -m gpu -x -q --timeout 800 -k "adv or ADV or refine" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["verified"]["exact"] if d.get("verified") else None)'
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
echo -n "rep $rep $lib --refine adv: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --refine adv 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
cd /tmp
for lib in libplslam_hip.so libplslam_hip_ab.so; do
PLSLAM_HIP_LIB=$ROOT/pl-slam_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/st_$lib" -o st -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify --refine adv > /dev/null 2>&1
f=$(find "$ROOT/$O/st_$lib" -name '*kernel_stats.csv' | head -1)
echo "== $lib" | tee -a "$ROOT/$O/ab.txt"
[ -n "$f" ] && grep -E "adv|rects" "$f" | cut -d, -f1-4,6 | tee -a "$ROOT/$O/ab.txt"
rm -rf "$ROOT/$O/st_$lib"
done
exit 0
