#!/bin/bash
# headline of several builds of the library alternating in one job: LIBS="libplslam_hip.so libplslam_hip_x.so ..." REPS=2 -> gpurun_out/abmany/ab.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/abmany
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; print(d["value"], d["ms_per_step"], {n: k[n] for n in k if "grow" in n or "rects" in n or "LBD" in n or "prep" in n})'
for rep in $(seq 1 ${REPS:-2}); do
for lib in ${LIBS:-libplslam_hip.so}; do
[ -f pl-slam_amd/$lib ] || continue
echo -n "rep $rep $lib: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-extras --no-verify ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
exit 0
