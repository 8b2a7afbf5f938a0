#!/bin/bash
# average latency of k_lsd_grow's memory operations (SQ_INST_LEVEL_* / SQ_INSTS_*), lone wavefronts (256 frames) vs full residency (6144)
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for B in 256 6144; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmclat$B -o o -- python $R/bench.py --steps 1 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > /dev/null 2>&1
  python - <<PY
import csv, collections
f="$R/gpurun_out/pmclat$B/o_counter_collection.csv"
acc=collections.defaultdict(float); n=0
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
    if k!="k_lsd_grow": continue
    acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": n+=1
print("batch $B launches",n,{c: round(v/n/$B) for c,v in acc.items()})
rd=acc["SQ_INSTS_VMEM_RD"]; wr=acc["SQ_INSTS_VMEM_WR"]
print("  avg VMEM latency (level/insts) %.0f ; LDS %.0f ; SMEM %.0f (counter units)"%(acc["SQ_INST_LEVEL_VMEM"]/max(rd+wr,1), acc["SQ_INST_LEVEL_LDS"]/max(acc["SQ_INSTS_LDS"],1), acc["SQ_INST_LEVEL_SMEM"]/max(acc["SQ_INSTS_SMEM"],1)))
PY
done
