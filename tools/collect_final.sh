#!/bin/bash
# copy what tools/gpu_final.sh left in gpurun_out/final/ into profiles/ under this round's names (ROUND=r06)
set -u
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
F=gpurun_out/final
cpn() { [ -s "$F/$1" ] && cp "$F/$1" "profiles/${R}_$2" && echo "profiles/${R}_$2"; }
cpn bench.json bench_full.json
cpn bench_force_dist.json bench_force_dist_one_rank.json
cpn kernel_stats_adv.csv full_pipeline_kernel_stats.csv
cpn kernel_stats_std.csv full_pipeline_kernel_stats_refine_std.csv
cpn kernel_stats_share512.csv share512_kernel_stats.csv
cpn kernel_stats_tracking.csv tracking_kernel_stats.csv
cpn residency_curve.txt residency_curve.txt
cpn tracking_bench.json tracking_bench_round_end.json
cpn adaptor_call_latency.txt adaptor_call_latency_round_end.txt
cpn kernel_resources.txt kernel_resources.txt
cpn pmc_insts.txt sq_instructions_per_kernel.txt
cpn pmc_grow_detail.txt grow_full_residency_counters.txt
cpn soak_1024.txt soak_parity_1024_frames.txt
{ echo "# pytest -m gpu + smoke on the round-end box (tools/gpu_final.sh), build $(python -c 'import __graft_entry__ as g; print(g.source_id())')"; cat $F/tests.txt $F/smoke.txt; } > profiles/${R}_gpu_tests.txt
[ -s $F/hbm_traffic.json ] && cp $F/hbm_traffic.json profiles/hbm_traffic.json
[ -s $F/insts.json ] && cp $F/insts.json profiles/sq_insts.json
[ -s $F/insts_tracking.json ] && cp $F/insts_tracking.json profiles/sq_insts_tracking.json
cpn pmc_tracking.txt sq_instructions_tracking.txt
exit 0
