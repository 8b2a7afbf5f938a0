#!/bin/bash
# one development job on the GPU box: GPU suite of the tree's build, PMC traffic of the tree's build, A/B of the tree's build against
# pl-slam_amd/libplslam_hip_ab.so (headline + kernels alone).  gpurun_out/job/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
if [ "${TESTS:-1}" = 1 ]; then
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 1200 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|rror|FAIL" | tail -8 | tee $O/tests.txt
fi
if [ "${PMC:-1}" = 1 ]; then
bash tools/pmc_traffic.sh 1536 > $O/pmc_traffic.log 2>&1; cp gpurun_out/pmc/traffic.json $O/traffic.json
python - <<'PY' | tee $O/traffic_summary.txt
import json
d=json.load(open('gpurun_out/job/traffic.json')); o=json.load(open('profiles/hbm_traffic.json'))
tot=0; toto=0
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['total']*kv[1].get('launches_per_step',1)):
    n=v.get('launches_per_step',1); ov=o['kernels'].get(k,{}); tot+=v['total']*n; toto+=ov.get('total',0)*ov.get('launches_per_step',1)
    if v['total']*n>20000: print('%-28s fetch %8.3f write %8.3f total %8.3f MB x%d   (was %8.3f)' % (k, v['fetch']/1e6, v['write']/1e6, v['total']/1e6, n, ov.get('total',0)/1e6))
print('total %.2f MB per frame (was %.2f)' % (tot/1e6, toto/1e6), d['build'], o['build'])
PY
fi
if [ "${INSTS:-0}" = 1 ]; then
bash tools/pmc_insts.sh 1536 > $O/pmc_insts.txt 2>&1; head -16 $O/pmc_insts.txt
fi
show='import json,sys
d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_launch"]; print(d["value"], d["ms_per_step"], k)'
for rep in 1 2; do
for lib in libplslam_hip.so libplslam_hip_ab.so; do
[ -f pl-slam_amd/$lib ] || continue
echo -n "rep $rep $lib headline: " | tee -a $O/ab.txt
PLSLAM_HIP_LIB=pl-slam_amd/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify 2>/dev/null | tail -1 | python -c "$show" | tee -a $O/ab.txt
done
done
exit 0
