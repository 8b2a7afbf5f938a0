#!/bin/bash
# the last run of a round: PMC traffic and instruction counts of THIS build -> profiles/ (on the box, so that bench.py's
# roofline.traffic / valu_issue carry a matching build id), then the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_traffic.sh 256 > $O/pmc_traffic.log 2>&1; tail -2 $O/pmc_traffic.log
bash tools/pmc_insts.sh 256 > $O/pmc_insts.txt 2>&1; head -4 $O/pmc_insts.txt
cp gpurun_out/pmc/traffic.json profiles/hbm_traffic.json
cp gpurun_out/pmcinst/insts.json profiles/r03_insts.json
cp gpurun_out/pmc/traffic.json $O/hbm_traffic.json
cp gpurun_out/pmcinst/insts.json $O/r03_insts.json
for c in fetch write; do f=$(find gpurun_out/pmc/$c -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python - "$f" "$O/r03_pmc_${c}_size_b256.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "counter", "launches", "sum"])
for (k, c), v in sorted(acc.items()): w.writerow([k, c, n[(k, c)], v])
PY
done
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'traffic', d['roofline'].get('traffic'), 'verified', d.get('verified',{}).get('exact'))
print('latency', d.get('latency_ms_single_frame'))
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
print('prov', d.get('pmc_provenance'))
PY
exit 0
