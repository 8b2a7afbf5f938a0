#!/bin/bash
# where do k_lsd_grow's cycles go at full residency (6144 frames in ONE launch = 6 wavefronts per SIMD)?  Two PMC passes.
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-6144}
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmcgrow$i -o o -- python $R/bench.py --steps 1 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > /dev/null 2>&1
done
python - <<PY
import csv, collections
for i in (1,2):
    f="$R/gpurun_out/pmcgrow%d/o_counter_collection.csv"%i
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("plh::","")
        if k not in ("k_lsd_grow","k_fast_strips","k_orient_brief"): continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
    for k in acc:
        print(k, "launches", n[k], {c: round(v/n[k]/$B) for c,v in acc[k].items()})
PY
