#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 gfx > gpurun_out/gpu.txt
lscpu | egrep 'Model name|^CPU\(s\)' >> gpurun_out/gpu.txt
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== bench KITTI 1241x376 / 2000 features (BASELINE.json configs[4] shape, one GPU's share)"
timeout 600 python bench.py --steps ${KITTI_STEPS:-3} --warmup 1 --rows 376 --cols 1241 --nfeatures 2000 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_kitti.log
echo "== rocprof"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o orb -- python "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/rocprof.log" 2>&1
cd "$OLDPWD"
find gpurun_out/prof -name '*kernel_stats*' | head -3
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); if [ -n "$f" ]; then head -12 "$f"; fi
exit 0
