#!/bin/bash
# SQ counters of the tracker's per-frame searches (tools/tracking_bench.py), wave-instructions per frame and kernel ->
# gpurun_out/pmctrack/insts.json (copied to profiles/sq_insts_tracking.json, which tracking_bench.py reads for its roofline block)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-1024}
S=2
rm -rf $R/gpurun_out/pmctrack
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmctrack -o o -- python $R/tools/tracking_bench.py --pairs $P --distinct 32 --steps $S --json > /dev/null 2>&1
python - <<PY
import csv, collections, json, sys
f="$R/gpurun_out/pmctrack/o_counter_collection.csv"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
    if not k.startswith("k_"): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
# launches of a kernel per step: k_proj_resolve runs 4 x, the prepasses 2 x each; steps = timed + warm-up + the verification pass
steps=max(n.get("k_frustum_points",1),1)
rows=[]
for k,a in acc.items():
    per=lambda c: a[c]/steps/$P
    rows.append((per("SQ_INSTS_VALU"), k, per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_VMEM"), per("SQ_WAVE_CYCLES"), per("SQ_WAIT_ANY"), n[k]/steps))
rows.sort(reverse=True)
print("%-26s %10s %10s %8s %8s %12s %12s %8s   (wave-instructions per frame; launches per step)"%("kernel","VALU","SALU","LDS","VMEM","wave_cycles","wait_any","launches"))
for v,k,s,l,m,wc,wa,nl in rows: print("%-26s %10.0f %10.0f %8.0f %8.0f %12.0f %12.0f %8.1f"%(k,v,s,l,m,wc,wa,nl))
sys.path.insert(0, "$R")
import __graft_entry__ as g
out={"build": g._lib_id(g.LIB), "what":"SQ counters per kernel of tools/tracking_bench.py --pairs $P, wave-instructions per frame (rocprofv3 --pmc, tools/pmc_tracking.sh)",
     "kernels":{k:{"valu":round(v),"salu":round(s),"lds":round(l),"vmem":round(m),"wave_cycles":round(wc),"wait_any":round(wa)} for v,k,s,l,m,wc,wa,nl in rows},
     "total_valu":round(sum(r[0] for r in rows)), "total_salu":round(sum(r[2] for r in rows)),
     "total_wave_cycles":round(sum(r[5] for r in rows)), "total_wait_any":round(sum(r[6] for r in rows))}
json.dump(out, open("$R/gpurun_out/pmctrack/insts.json","w"), indent=1)
PY
