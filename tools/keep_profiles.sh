#!/bin/bash
# Here (not on the GPU box): copy what tools/collect_profiles.sh left under gpurun_out/profile_<tag>/ into profiles/<tag>_*,
# under the names profiles/README.md describes.
set -eu
TAG=${1:-r04}
S=gpurun_out/profile_$TAG
D=profiles
for f in bench bench_driver_args bench_four_launches bench_three_launches bench_2ranks_one_gpu_gloo pipelined_kernel_us pmc_summary; do cp $S/$f.json $D/${TAG}_$f.json; done
cp $S/bench_cold_250.json $D/${TAG}_bench_cold_250_steps.json
for f in step_spans step_timeline step_timeline_three_launches emit_phases overlap_phases rows_phases shard_phases pcie_rate soak; do cp $S/$f.txt $D/${TAG}_$f.txt; done
for f in shard_rehearsal shard_rehearsal_large scan_stress scan_stride; do cp $S/$f.jsonl $D/${TAG}_$f.jsonl; done
for f in timeline_large row_lengths_large pool_growth; do cp $S/$f.txt $D/${TAG}_$f.txt; done
newest() { ls -t $1 | head -1; }       # (gpurun merges every call's files into gpurun_out/: earlier collections' are still there)
cp $(newest "$S/stats_pipelined/*/*_kernel_stats.csv") $D/${TAG}_kernel_stats_pipelined.csv
cp $(newest "$S/stats/*/*_kernel_stats.csv") $D/${TAG}_kernel_stats_one_role_per_launch.csv
cp $(newest "$S/stats_stress/*/*_kernel_stats.csv") $D/${TAG}_kernel_stats_configs4_leg.csv
cp $(newest "$S/stats_large/*/*_kernel_stats.csv") $D/${TAG}_kernel_stats_large_pool.csv
cp $(newest "$S/stats_pipelined/*/*_agent_info.csv") $D/${TAG}_agent_info.csv
