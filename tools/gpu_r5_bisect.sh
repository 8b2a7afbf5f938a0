#!/bin/bash
# Round 5, job 2: where did round 4 lose 5 %?  Every round-4 kernel commit as its own tree (ab_bis/<sha>, built from `git archive`),
# round 3's final tree (ab_r3) first and last, two passes, one job, one box.  gpurun_out/r5bis/bisect.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5bis
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; t=d.get("kernel_ms_per_launch_timed_region") or {}
short=lambda n: n.split(" ")[0] if not n.startswith("line") and not n.startswith("LBD") else n[:4]
print("%8.0f f/s %7.2f ms | alone" % (d["value"], d["ms_per_step"]), " ".join("%s %.2f" % (short(n), k[n]) for n in k), "| timed", " ".join("%s %.1f" % (short(n), t[n]) for n in t))'
args="--steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify"
for rep in 1 2; do
  for t in ab_r3 ${TREES:-ab_bis/13a58f9 ab_bis/7c7701e ab_bis/96ab0f5 ab_bis/7f3b6b8 ab_bis/dfe777f ab_bis/ae8e23c ab_bis/dd7c5e2 ab_bis/a9090a0 ab_bis/4cf652c} ab_r3; do
    printf "rep $rep %-16s: " $t | tee -a $O/bisect.txt
    (cd $t && timeout 300 python bench.py $args 2>$OLDPWD/$O/err.txt | tail -1 | python -c "$show" 2>&1 | tail -1) | tee -a $O/bisect.txt
  done
done
exit 0
