#!/bin/bash
# round 4, first box: parity of the deferred-rectangle design (line tests, soak with the screen's self-check), A/B of the density
# screen on the headline, the default bench.  Everything lands in gpurun_out/r4a/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_abi.py tests/test_line.py tests/test_e2e_gpu.py -m gpu -x -q --timeout 600 2>&1 | tail -6 | tee $O/tests_line.txt
timeout 1200 python -m pytest tests/test_soak_gpu.py -m gpu -x -q -s --timeout 1000 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/tests_soak.txt
for rep in 1 2; do
for flag in "" "--no-screen"; do
echo -n "rep $rep screen[$flag] headline: " | tee -a $O/ab_screen.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-verify $flag 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])" | tee -a $O/ab_screen.txt
done
done
echo -n "adv headline: " | tee -a $O/ab_screen.txt
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --refine adv 2>$O/adv.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_launch'], d['verified']['exact'])" | tee -a $O/ab_screen.txt
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4a/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['ms_per_launch'], d['roofline'].get('ms_per_launch_alone'), 'verified', d.get('verified',{}).get('exact'), d.get('verified',{}).get('frames'))
print('latency', d.get('latency_ms_single_frame'))
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
a=s.get('refine_adv',{})
print('adv', a.get('value'), a.get('vs_headline'), 'share', a.get('share_512',{}).get('value'), 'lat', a.get('latency_ms_single_frame'), 'ver', (a.get('verified') or {}).get('exact'))
print('streaming', d.get('streaming',{}).get('value'), 'cpu', d.get('cpu_baseline'))
print('extras_seconds', d.get('extras_seconds'))
PY
exit 0
