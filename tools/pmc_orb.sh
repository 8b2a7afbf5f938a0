cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcorb -o o -- python $R/tools/orb_time.py > /dev/null 2>&1
python - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
f=R+"/gpurun_out/pmcorb/o_counter_collection.csv"
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k in acc:
    if "plh" in k:
        print(k, n[k], {c: round(v/max(n[k],1)) for c,v in acc[k].items()})
PY
