#!/bin/bash
# A/B of a variant build (LIB=libplslam_hip_dpp.so) against libplslam_hip_ab.so: selftest + line / e2e GPU tests of the variant, then the
# headline of both alternating, then rocprofv3 statistics of k_lsd_grow per build.  gpurun_out/abdpp/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/abdpp
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
V=${LIB:-libplslam_hip_dpp.so}
PLSLAM_HIP_LIB=$ROOT/pl-slam_amd/$V timeout 1200 python -m pytest tests/test_abi.py tests/test_line.py tests/test_e2e_gpu.py -m gpu -x -q --timeout 800 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; print(d["value"], d["ms_per_step"], (d.get("verified") or {}).get("exact"), {n: k[n] for n in k if "grow" in n})'
for rep in 1 2 3; do
for lib in $V libplslam_hip_ab.so; do
echo -n "rep $rep $lib: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=$ROOT/pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
cd /tmp
for lib in $V libplslam_hip_ab.so; do
PLSLAM_HIP_LIB=$ROOT/pl-slam_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/st_$lib" -o st -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify > /dev/null 2>&1
f=$(find "$ROOT/$O/st_$lib" -name '*kernel_stats.csv' | head -1)
echo "== $lib" | tee -a "$ROOT/$O/ab.txt"
[ -n "$f" ] && grep -E "grow|rects" "$f" | cut -d, -f1-4,6 | tee -a "$ROOT/$O/ab.txt"
rm -rf "$ROOT/$O/st_$lib"
done
exit 0
