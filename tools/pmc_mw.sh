#!/bin/bash
# SQ counters of k_lsd_grow_mw / k_lsd_grow_lone for a small batch: pmc_mw.sh BATCH "WAVES..." (three PMC passes each)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
P3="SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_IFETCH"
for W in ${2:-0 8}; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmcmw_${W}_$i
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmcmw_${W}_$i -o o -- python $R/tools/mw_one.py $B $W 4 > /dev/null 2>&1
  done
  python - <<PY
import csv, collections, glob
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for i in (1,2,3):
    for f in glob.glob("$R/gpurun_out/pmcmw_${W}_%d/**/*counter_collection.csv"%i, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
            if not k.startswith("k_lsd_grow"): continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    print("waves $W batch $B", k, {c: round(v/n[k][c]/$B) for c,v in sorted(acc[k].items())})
PY
done
