#!/bin/bash
# round-end evidence on one box: GPU suite, smoke, default bench; kernel stats of the bench; PMC traffic and instruction counts;
# multi-wavefront sweep and counters.  Everything lands in gpurun_out/final/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
O=gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline'], 'verified', d.get('verified',{}).get('exact'))
print('latency', d.get('latency_ms_single_frame'))
s=d.get('secondary',{})
print('secondary', s.get('value'), 'share512', s.get('configs4_share_512',{}).get('value'), s.get('error'))
print('streaming', d.get('streaming',{}).get('value'), 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/stats" -o st -- \
  python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify > "$ROOT/$O/bench_stats.log" 2>&1
cd "$ROOT"
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -8 "$f"
bash tools/pmc_traffic.sh 256 > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
bash tools/pmc_insts.sh 256 > $O/pmc_insts.txt 2>&1; head -8 $O/pmc_insts.txt
timeout 600 python tools/mw_sweep.py --batches 1,8,512,1024 --waves 0,4,8,16 --reps 4 2>&1 | grep -v amdgpu.ids | tee $O/mw_sweep.txt
bash tools/pmc_mw.sh 1 "0 8 16" 2>&1 | grep -v amdgpu.ids | tee $O/pmc_mw.txt
exit 0
