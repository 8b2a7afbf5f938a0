#!/bin/bash
# Round 5: variants of the library (pl-slam_amd/libplslam_hip_$v.so; "tree" = the tree's build): per-kernel durations alone at 1536 frames
# per launch (rocprofv3 kernel trace of bench.py --serial) for the kernels matching $KERNELS, and the pipelined rate.  gpurun_out/r5var/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=$ROOT/gpurun_out/r5var
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
KERNELS=${KERNELS:-"k_lsd_rects|k_adv_|k_keylines"}
LVL=${LVL:-adv}
for v in ${VARS:-tree}; do
  f=$ROOT/pl-slam_amd/libplslam_hip.so; [ "$v" != tree ] && f=$ROOT/pl-slam_amd/libplslam_hip_$v.so
  [ "${ALONE:-1}" = 1 ] && (cd /tmp && PLSLAM_HIP_LIB=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o s -- python $ROOT/bench.py --refine ${LVL%% *} --serial --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/$v.log 2>&1)
  [ "${ALONE:-1}" = 1 ] && python - <<PY | tee -a $O/variants.txt
import csv, glob, re
rows=[]
for f in glob.glob("$O/$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0].replace("plh::","").replace("void ","")
        if re.search(r"$KERNELS", n): rows.append("%s %.3f" % (n, float(r["AverageNs"])/1e6))
print("%-8s alone (ms per 1536 frames): %s" % ("$v", "  ".join(sorted(rows))))
PY
  if [ "${RATE:-1}" = 1 ]; then
  for L in $LVL; do
  printf "%-8s pipelined %s: " $v $L | tee -a $O/variants.txt
  PLSLAM_HIP_LIB=$f timeout 600 python bench.py --refine $L --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f f/s %.2f ms  box %s" % (d["value"], d["ms_per_step"], (d.get("box") or {}).get("probe_ms")))' | tee -a $O/variants.txt
  done
  fi
done
exit 0
