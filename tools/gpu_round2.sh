#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest match+line gpu"
timeout 900 python -m pytest tests/test_match.py tests/test_line.py -m gpu -x -q -s --timeout 300 2>&1 | tail -30 | tee gpurun_out/pytest_line_match.log
echo "== line bench"
timeout 300 python tools/line_bench.py 256 2>&1 | tail -5 | tee gpurun_out/line_bench.log
echo "== rocprof line"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_line" -o line -- python "$OLDPWD/tools/line_bench.py" 256 > "$OLDPWD/gpurun_out/rocprof_line.log" 2>&1
cd "$OLDPWD"
f=$(find gpurun_out/prof_line -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then cat "$f" | cut -c1-200 | head -20; fi
exit 0
