#!/bin/bash
# Instruction-issue budget per kernel (SQ counters, one pass): which kernels own the SIMD issue slots.
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-256}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcinst -o o -- python $R/bench.py --steps 2 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > /dev/null 2>&1
python - <<PY
import csv, collections
f="$R/gpurun_out/pmcinst/o_counter_collection.csv"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")   # template kernels print as "void plh::k<..>(...)"
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
steps=max(n.get("k_lsd_grow",1),1)
rows=[]
for k in acc:
    if k.startswith("k_") and k != "k_box_probe":   # (bench.py's box normaliser is not part of the front end)
        a=acc[k]; per=lambda c: a[c]/steps/$B
        rows.append((per("SQ_INSTS_VALU")+per("SQ_INSTS_SALU")+per("SQ_INSTS_LDS")+per("SQ_INSTS_VMEM"), k, per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM"), per("SQ_WAVE_CYCLES"), per("SQ_WAIT_ANY")))
rows.sort(reverse=True)
print("%-18s %12s %12s %10s %10s %14s %14s   (wave-instructions per frame)"%("kernel","VALU","SALU","LDS","VMEM","wave_cycles","wait_any"))
for t,k,v,s,l,m,wc,wa in rows: print("%-18s %12.0f %12.0f %10.0f %10.0f %14.0f %14.0f"%(k,v,s,l,m,wc,wa))
import json, subprocess, sys
sys.path.insert(0, "$R")
import __graft_entry__ as g
out={"build": g._lib_id(g.LIB), "what":"SQ counters per kernel, wave-instructions per frame (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY, tools/pmc_insts.sh $B: bench.py --batch $B --nsplit 1 --serial, 640x480 / 1000 ORB / 200 lines)",
     "kernels":{k:{"valu":round(v),"salu":round(s),"lds":round(l),"vmem":round(m),"wave_cycles":round(wc),"wait_any":round(wa)} for t,k,v,s,l,m,wc,wa in rows},
     "total_valu":round(sum(r[2] for r in rows)), "total_salu":round(sum(r[3] for r in rows))}
json.dump(out, open("$R/gpurun_out/pmcinst/insts.json","w"), indent=1)
PY
