#!/bin/bash
# Final measurement round of a build: GPU parity suite, smoke, bench (default flags = what the driver runs), rocprofv3 kernel
# stats of the same command, SQ instruction counters, HBM traffic counters (two PMC passes).  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 gfx > gpurun_out/gpu.txt
lscpu | egrep 'Model name|^CPU\(s\)' >> gpurun_out/gpu.txt
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v "^  \|^$" | tail -12 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== bench (default flags)"
( time timeout 1200 python bench.py ) 2>&1 | tail -5 > gpurun_out/bench_full.log; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_full.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
print(d['value'], 'frames/s;', 'roofline', d['roofline']['frac'], 'fast', d['roofline_fast']['frac'], d['roofline_fast'].get('valu_issue',{}).get('frac'))
print('streaming', d.get('streaming',{}).get('value'), 'latency', d.get('latency_ms_single_frame'))
s=d.get('secondary',{}); print('secondary', s.get('value'), s.get('configs4_share_512',{}).get('value'), s.get('error'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
tail -4 gpurun_out/bench_full.log | grep real
echo "== bench --force-dist (1-rank RCCL rehearsal of the N > 1 path)"
timeout 600 python bench.py --force-dist --no-cpu-baseline --no-extras --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('rccl'))" 2>&1 | cut -c1-300
echo "== rocprof kernel stats of bench.py (5 steps)"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_final" -o full -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OLDPWD/gpurun_out/rocprof_final.log" 2>&1
cd "$OLDPWD"
f=$(find gpurun_out/prof_final -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then grep -v "at::native\|rocclr" "$f" | cut -c1-150 | head -14; fi
echo "== SQ instruction counters"
timeout 900 bash tools/pmc_insts.sh 256 2>&1 | tail -22 | tee gpurun_out/pmc_insts.log
echo "== HBM traffic counters"
timeout 1200 bash tools/pmc_traffic.sh 256 > gpurun_out/pmc_traffic.log 2>&1; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/pmc/traffic.json'))
    for k in ('k_pyr_down','k_fast_strips','k_lsd_grow','k_orient_brief','k_lsd_grad'):
        print(k, d['kernels'].get(k))
except Exception as e: print('traffic failed', e)
PY
exit 0
