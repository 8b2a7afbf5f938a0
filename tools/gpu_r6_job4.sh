#!/bin/bash
# Round 6, job 4: the projection searches as prepass + ordered resolve (parity, adaptor call latency, resident-batch throughput), and which
# frames of the 1024-frame soak the counter build gets wrong.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job4
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_search.py tests/test_ref_track.py tests/test_adaptor_exec.py tests/test_frustum.py -m gpu -x -q --timeout 600 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_search.txt
timeout 900 python tools/adaptor_latency.py 30 > $O/adaptor_call_latency.txt 2> $O/adaptor_call_latency.err; echo "latency exit $?"; cat $O/adaptor_call_latency.txt; tail -3 $O/adaptor_call_latency.err
timeout 900 python tools/tracking_bench.py --json > $O/tracking_bench.json 2> $O/tracking_bench.err; echo "tracking exit $?"; cat $O/tracking_bench.json; tail -5 $O/tracking_bench.err
PLSLAM_SOAK_FRAMES=1024 timeout 900 python -m pytest tests/test_soak_gpu.py -m gpu -q -s --timeout 800 -k "distinct and 640x480 and adv" 2>&1 | grep -v amdgpu.ids | grep -E "^soak|passed|failed|Error|error|counter build" | tee $O/soak1024_640_adv.txt
exit 0
