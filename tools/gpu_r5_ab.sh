#!/bin/bash
# Round 5, job 1: round 3's final build (its own tree under ab_r3/, commit a95f811) against round 4's (HEAD's kernels:
# libplslam_hip_r4.so) and the round-5 candidates, alternating in ONE job on ONE box.  gpurun_out/r5ab/ab.txt
# Variants: LIBS="r4 split w8 ''" ('' = the tree's default build).  REPS alternations.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5ab
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
REPS=${REPS:-3}
LIBS=${LIBS:-"r4 split default w8"}
rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -v "^$" | head -40 > $O/smi_before.txt
if [ "${TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 800 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.txt
fi
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; t=d.get("kernel_ms_per_launch_timed_region") or {}
short=lambda n: n.split(" ")[0] if not n.startswith("line") and not n.startswith("LBD") else n[:4]
print("%8.0f f/s %7.2f ms | alone" % (d["value"], d["ms_per_step"]), " ".join("%s %.2f" % (short(n), k[n]) for n in k), "| timed", " ".join("%s %.1f" % (short(n), t[n]) for n in t))'
args="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
for rep in $(seq 1 $REPS); do
  echo -n "rep $rep r3      : " | tee -a $O/ab.txt
  (cd ab_r3 && timeout 600 python bench.py $args 2>/dev/null | tail -1 | python -c "$show") | tee -a $O/ab.txt
  L="$LIBS"; [ $rep -gt 2 ] && L="${LIBS_LATE:-$LIBS}"
  for lib in $L; do
    f=pl-slam_amd/libplslam_hip_$lib.so; [ "$lib" = default ] && f=pl-slam_amd/libplslam_hip.so
    printf "rep $rep %-8s: " $lib | tee -a $O/ab.txt
    PLSLAM_HIP_LIB=$f timeout 600 python bench.py $args 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
  done
done
if [ "${HALVES:-1}" = 1 ]; then
  echo "halves r3:" | tee -a $O/ab.txt; (cd ab_r3 && timeout 300 python tools/halves.py 2>/dev/null) | tee -a $O/ab.txt
  for lib in r4 default; do
    f=pl-slam_amd/libplslam_hip_$lib.so; [ "$lib" = default ] && f=pl-slam_amd/libplslam_hip.so
    echo "halves $lib:" | tee -a $O/ab.txt; PLSLAM_HIP_LIB=$f timeout 300 python tools/halves.py 2>/dev/null | tee -a $O/ab.txt
  done
fi
if [ "${SHARE:-1}" = 1 ]; then
  for lib in r4 default; do
    f=pl-slam_amd/libplslam_hip_$lib.so; [ "$lib" = default ] && f=pl-slam_amd/libplslam_hip.so
    printf "share512 %-8s: " $lib | tee -a $O/ab.txt
    PLSLAM_HIP_LIB=$f timeout 600 python bench.py --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
    printf "adv      %-8s: " $lib | tee -a $O/ab.txt
    PLSLAM_HIP_LIB=$f timeout 600 python bench.py --refine adv $args 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
  done
fi
rocm-smi --showclocks --showpower 2>/dev/null | grep -v "^$" | head -30 > $O/smi_after.txt
exit 0
