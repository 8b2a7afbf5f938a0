#!/bin/bash
# Round 5: quick check of a change to the line half -- line tests, then ADV / STD rates of the tree (and of round 4's tree once), the
# configs[4] share, the instruction table.  gpurun_out/r5j7/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5j7
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest ${TESTSEL:-tests/test_line.py tests/test_e2e_gpu.py tests/test_abi.py} -m gpu -x -q --timeout 1200 2>&1 | grep -v "amdgpu.ids\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | tee $O/tests.txt
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; t=d.get("kernel_ms_per_launch_timed_region") or {}
short=lambda n: n.split(" ")[0] if not n.startswith("line") and not n.startswith("LBD") else n[:4]
print("%8.0f f/s %7.2f ms | alone" % (d["value"], d["ms_per_step"]), " ".join("%s %.2f" % (short(n), k[n]) for n in k), "| timed", " ".join("%s %.1f" % (short(n), t[n]) for n in t), "| box", (d.get("box") or {}).get("probe_ms"))'
args="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
printf "r4   adv : " | tee -a $O/ab.txt
(cd ab_r4 && timeout 600 python bench.py --refine adv $args 2>/dev/null | tail -1 | python -c "$show") | tee -a $O/ab.txt
for rep in 1 2; do
  for lvl in adv std; do
    printf "rep $rep tree $lvl : " | tee -a $O/ab.txt
    timeout 600 python bench.py --refine $lvl $args 2>$O/err_$lvl.txt | tail -1 | python -c "$show" | tee -a $O/ab.txt
  done
done
printf "share512 adv tree: " | tee -a $O/ab.txt
timeout 600 python bench.py --refine adv --batch 512 --nsplit 1 --rows 376 --cols 1241 --nfeatures 2000 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
bash tools/pmc_insts.sh 256 > $O/pmcinst.txt 2>&1
exit 0
