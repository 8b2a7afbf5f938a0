#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -2 | tee gpurun_out/bench.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_full" -o full -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/rocprof_full.log" 2>&1
cd "$OLDPWD"
f=$(find gpurun_out/prof_full -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then grep -v "at::native\|rocclr" "$f" | cut -c1-160 | head -24; fi
exit 0
