#!/bin/bash
# LDS bank conflicts per kernel: SQ_LDS_BANK_CONFLICT (cycles lost) against SQ_LDS_IDX_ACTIVE (cycles the LDS index unit is busy)
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-256}
rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmclds -o o -- python $R/bench.py --steps 2 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > /dev/null 2>&1
python - <<PY
import csv, collections
f="$R/gpurun_out/pmclds/o_counter_collection.csv"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
steps=max(n.get("k_lsd_grow",1),1)
print("%-20s %12s %12s %8s %12s %12s %12s"%("kernel","bank_confl","idx_active","ratio","addr_confl","unaligned","lds_insts"))
rows=[]
for k,a in acc.items():
    if not k.startswith("k_"): continue
    p=lambda c: a[c]/steps/$B
    rows.append((p("SQ_LDS_BANK_CONFLICT"),k,p("SQ_LDS_IDX_ACTIVE"),p("SQ_LDS_ADDR_CONFLICT"),p("SQ_LDS_UNALIGNED_STALL"),p("SQ_INSTS_LDS")))
for bc,k,ia,ac,ua,li in sorted(rows,reverse=True):
    print("%-20s %12.0f %12.0f %8.2f %12.0f %12.0f %12.0f"%(k,bc,ia,bc/max(ia,1),ac,ua,li))
PY
