#!/bin/bash
# Round 5: the tree's build against a variant library (pl-slam_amd/libplslam_hip_$VAR.so), alternating; ADV and STD.  gpurun_out/r5j8/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5j8
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
VAR=${VAR:-cap}
[ "${TESTS:-1}" = 1 ] && timeout 1500 python -m pytest ${TESTSEL:-tests/test_line.py} -m gpu -x -q --timeout 1200 2>&1 | grep -v "amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; t=d.get("kernel_ms_per_launch_timed_region") or {}
short=lambda n: n.split(" ")[0] if not n.startswith("line") and not n.startswith("LBD") else n[:4]
print("%8.0f f/s %7.2f ms | alone" % (d["value"], d["ms_per_step"]), " ".join("%s %.2f" % (short(n), k[n]) for n in k), "| timed", " ".join("%s %.1f" % (short(n), t[n]) for n in t), "| box", (d.get("box") or {}).get("probe_ms"))'
args="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
for rep in 1 2 ${REPS3:-}; do
  for lvl in ${LEVELS:-adv std}; do
    printf "rep $rep tree $lvl : " | tee -a $O/ab.txt
    timeout 600 python bench.py --refine $lvl $args 2>$O/err_$lvl.txt | tail -1 | python -c "$show" | tee -a $O/ab.txt
    for v in $VAR; do
    printf "rep $rep %-4s $lvl : " $v | tee -a $O/ab.txt
    PLSLAM_HIP_LIB=pl-slam_amd/libplslam_hip_$v.so timeout 600 python bench.py --refine $lvl $args 2>$O/err_${v}_$lvl.txt | tail -1 | python -c "$show" | tee -a $O/ab.txt
    done
  done
done
if [ "${SHARE:-1}" = 1 ]; then
for v in "" $VAR; do
  f=pl-slam_amd/libplslam_hip.so; [ -n "$v" ] && f=pl-slam_amd/libplslam_hip_$v.so
  printf "share512 adv %-5s: " "${v:-tree}" | tee -a $O/ab.txt
  PLSLAM_HIP_LIB=$f timeout 600 python bench.py --refine adv --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
fi
[ "${PMC:-0}" = 1 ] && bash tools/pmc_insts.sh 256 > $O/pmcinst.txt 2>&1
exit 0
