#!/bin/bash
# after a scheduling change: line tests, bench, kernel-trace timeline of the overlapped run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "line or e2e or adaptor" 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/bench_r14.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r14.json'))
print(d['value'], 'frames/s;', d['ms_per_step'], 'ms/step; roofline', d['roofline']['frac'], 'fast', d['roofline_fast']['frac'])
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_r14" -o full -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OLDPWD/gpurun_out/rocprof_r14.log" 2>&1
cd "$OLDPWD"
f=$(find gpurun_out/prof_r14 -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then grep -v "at::native\|rocclr" "$f" | cut -c1-110 | head -16; fi
exit 0
