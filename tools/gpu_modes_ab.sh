#!/bin/bash
# A/B of the modes of one build: A/B (screen on / off / ADV), the default bench line, PMC calibration of the counters' access patterns.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/modes
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for flag in "" "--no-screen" "--refine adv"; do
echo -n "[$flag] headline: " | tee -a $O/ab.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify $flag 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/ab.txt
done
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/modes/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('ms_per_launch_alone'), 'verified', d.get('verified',{}).get('exact'), d.get('verified',{}).get('frames'))
print('latency', {k: v for k, v in d.get('latency_ms_single_frame', {}).items() if k != 'note'})
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
a=s.get('refine_adv',{})
print('adv', a.get('value'), a.get('vs_headline'), 'share', a.get('share_512',{}).get('value'), 'lat', {k: v for k, v in (a.get('latency_ms_single_frame') or {}).items() if k != 'note'}, 'ver', (a.get('verified') or {}).get('exact'))
c=d.get('cpu_baseline',{})
print('streaming', d.get('streaming',{}).get('value'), 'cpu', c.get('value'), c.get('cores'), c.get('legs'))
print('extras_seconds', d.get('extras_seconds'))
PY
bash tools/pmc_calibrate.sh 2048 2>&1 | tail -40 | tee $O/pmc_cal.txt
cp gpurun_out/pmc_cal/calibration.json $O/ 2>/dev/null
exit 0
