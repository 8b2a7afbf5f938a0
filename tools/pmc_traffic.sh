#!/bin/bash
# HBM traffic of the front-end kernels from the L2 memory-side counters (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 TCC slots of 4) -> two separate --pmc runs, kernel trace only.
# Run on the GPU box:  bash tools/pmc_traffic.sh [batch]   -> gpurun_out/pmc/{fetch,write}/... + gpurun_out/pmc/traffic.json
GW="--grow-waves 0"   # these profiles are about the one-wavefront-per-frame kernels (small batches would run k_lsd_grow_mw)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$PWD
B=${1:-256}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/gpurun_out/pmc/$d" -o pmc -- \
    python "$ROOT/bench.py" --steps 2 --warmup 1 --batch $B --nsplit 1 --no-cpu-baseline --no-extras --no-verify --serial $GW > "$ROOT/gpurun_out/pmc/$d.log" 2>&1
done
cd "$ROOT"
python tools/pmc_summarize.py gpurun_out/pmc $B | tee gpurun_out/pmc/traffic.json
