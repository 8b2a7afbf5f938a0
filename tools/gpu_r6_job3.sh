#!/bin/bash
# Round 6, job 3: the counter build after the fix, adaptor call latency (reference CPU code vs drop-in on the GPU, same objects), the
# tracker's per-frame searches as a resident batch, the GPU suite, the 1024-frame soak on this build.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/job3
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for c in "300 -1" "1024 -1" "256 16" "1024 4"; do
  timeout -s KILL 150 python tools/dbg/fault_case.py libplslam_hip_prof.so $c > $O/prof_case.out 2> $O/prof_case.err
  echo "prof $c: exit $? $(grep -h 'status\|Memory access' $O/prof_case.out $O/prof_case.err | tail -1)" | tee -a $O/prof_cases.txt
done
timeout 900 python tools/adaptor_latency.py 30 > $O/adaptor_call_latency.txt 2> $O/adaptor_call_latency.err; echo "latency exit $?"; cat $O/adaptor_call_latency.txt; tail -3 $O/adaptor_call_latency.err
timeout 900 python tools/tracking_bench.py --json > $O/tracking_bench.json 2> $O/tracking_bench.err; echo "tracking exit $?"; cat $O/tracking_bench.json; tail -5 $O/tracking_bench.err
timeout 1800 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.txt
PLSLAM_SOAK_FRAMES=1024 timeout 1500 python -m pytest tests/test_soak_gpu.py -m gpu -q -s --timeout 1200 2>&1 | grep -v amdgpu.ids | grep -E "^soak|passed|failed|Error|error" | tee $O/soak1024.txt
exit 0
