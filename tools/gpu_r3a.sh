#!/bin/bash
# adaptor back-end tests on the GPU + kernel stats of the default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_adaptor_exec.py -m gpu -x -q --timeout 600 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r3a/adaptor.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r3a/stats" -o st -- \
  python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-verify > "$ROOT/gpurun_out/r3a/bench_stats.log" 2>&1
cd "$ROOT"
f=$(find gpurun_out/r3a/stats -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r3a/kernel_stats.csv && head -30 "$f"
tail -1 gpurun_out/r3a/bench_stats.log | cut -c1-600
exit 0
