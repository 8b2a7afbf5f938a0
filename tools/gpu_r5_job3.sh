#!/bin/bash
# Round 5, job 3: GPU suite on the tree's build (LSD_REFINE_ADV default, fused ADV kernels), then A/B against round 4's final
# tree (ab_r4 = 7c3b187) at both refine levels, alternating, one job.  gpurun_out/r5j3/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5j3
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
if [ "${TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest ${TESTSEL:-tests} -m gpu -x -q --timeout 1200 2>&1 | grep -v "amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 | tee $O/tests.txt
fi
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; t=d.get("kernel_ms_per_launch_timed_region") or {}
short=lambda n: n.split(" ")[0] if not n.startswith("line") and not n.startswith("LBD") else n[:4]
print("%8.0f f/s %7.2f ms | alone" % (d["value"], d["ms_per_step"]), " ".join("%s %.2f" % (short(n), k[n]) for n in k), "| timed", " ".join("%s %.1f" % (short(n), t[n]) for n in t))'
args="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
for rep in 1 2; do
  for lvl in adv std; do
    printf "rep $rep r4   $lvl : " | tee -a $O/ab.txt
    (cd ab_r4 && timeout 600 python bench.py --refine $lvl $args 2>/dev/null | tail -1 | python -c "$show") | tee -a $O/ab.txt
    printf "rep $rep tree $lvl : " | tee -a $O/ab.txt
    timeout 600 python bench.py --refine $lvl $args 2>$O/err_$lvl.txt | tail -1 | python -c "$show" | tee -a $O/ab.txt
  done
done
printf "tree adv verified: " | tee -a $O/ab.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>$O/err_verify.txt | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["lsd_refine"]["level"], d["verified"])' | tee -a $O/ab.txt
for t in ab_r4 .; do
  printf "share512 adv %-6s: " $t | tee -a $O/ab.txt
  (cd $t && timeout 600 python bench.py --refine adv --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show") | tee -a $O/ab.txt
done
bash tools/pmc_insts.sh 256 > $O/pmcinst.txt 2>&1
exit 0
