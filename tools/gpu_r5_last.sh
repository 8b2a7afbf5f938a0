#!/bin/bash
# Round 5, last job: GPU suite + smoke on the final tree, the default bench line (PMC profiles of this build are in profiles/), the N > 1 code
# path on one GPU (--force-dist: a one-rank RCCL communicator, weak and strong).  gpurun_out/last/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/last
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee $O/smoke.txt
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY' | tee $O/bench.txt
import json
d=json.load(open('gpurun_out/last/bench.json'))
print('value', d['value'], d['config']['lsd_refine']['level'], 'box', (d.get('box') or {}).get('probe_ms'), 'ms/step', d['ms_per_step'], 'verified', d['verified']['exact'], d['verified']['frames'])
print('roofline', {k: d['roofline'][k] for k in ('frac','achieved','traffic','ms_per_launch','ms_per_launch_alone')}, 'valu', d['roofline'].get('valu_issue',{}).get('frac'), 'front end', d['front_end_valu_issue']['frac'], d['front_end_valu_issue']['valu_wave_instructions_per_frame'])
print('latency', {k: v for k, v in d.get('latency_ms_single_frame', {}).items() if k != 'note'})
s=d.get('secondary',{}); a=s.get('refine_std',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), '| STD', a.get('value'), a.get('vs_headline'), 'share', a.get('share_512',{}).get('value'))
c=d.get('cpu_baseline',{}); print('streaming', d.get('streaming',{}).get('value'), 'cpu', c.get('value'), c.get('cores'))
PY
printf "force-dist weak   : " | tee -a $O/dist.txt
timeout 600 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["scaling"], d["verified"]["exact"], (d.get("rccl") or {}).get("version"))' | tee -a $O/dist.txt
printf "force-dist strong : " | tee -a $O/dist.txt
timeout 600 python bench.py --force-dist --scaling strong --total 512 --rows 376 --cols 1241 --nfeatures 2000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["scaling"], d["verified"]["exact"], d["config"]["workload"][:60])' | tee -a $O/dist.txt
exit 0
