#!/bin/bash
# memory-side view of k_lsd_grow, lone wavefronts (256 frames) vs full residency (6144): L1->L2 read latency, address translation, L2 hit rate
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PA="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum"
PB="TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
for B in 256 6144; do
 for P in "$PA" "$PB"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmcmem -o o -- python $R/bench.py --steps 1 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --serial $GW > /dev/null 2>&1
  python - <<PY
import csv, collections
f="$R/gpurun_out/pmcmem/o_counter_collection.csv"
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
    if k!="k_lsd_grow": continue
    acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("batch $B", {c: round(v/max(n[c],1)/$B) for c,v in acc.items()})
PY
 done
done
