#!/bin/bash
# what does the vector memory path do under k_lsd_grow at full residency (6144 frames in ONE launch)?  TA / TCP / UTCL1 / TCC counters,
# two counters of one block per pass (the profiler refuses six TCP / TA / TCC counters at once -- and its child then hangs: hence the
# short timeouts); stops after two refused passes.  Output: gpurun_out/pmcmem/summary.txt
GW="--grow-waves 0"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-6144}
O=$R/gpurun_out/pmcmem
rm -rf $O; mkdir -p $O
PASSES=(
"TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
"TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"
"TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum"
"TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
"TCC_HIT_sum TCC_MISS_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
"TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
"GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
)
i=0; fails=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  [ "$fails" -ge 2 ] && break
  timeout -k 5 100 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o o -- python $R/bench.py --steps 1 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > $O/p$i.log 2>&1 || { echo "pass $i failed: $P" >> $O/summary.txt; fails=$((fails+1)); }
done
python - <<PY >> $O/summary.txt
import csv, collections, glob
for i in range(1, $i + 1):
    fs = glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("pass", i, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    first = None
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("plh::", "")
        if k not in ("k_lsd_grow", "k_lsd_rects", "k_lsd_bin_scatter"): continue
        if first is None: first = r["Counter_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == first: n[k] += 1
    for k in acc:
        print(k, "launches", n[k], "per frame:", {c: round(v / max(n[k], 1) / $B, 1) for c, v in acc[k].items()})
PY
cat $O/summary.txt
rm -rf $O/p[0-9]*/
