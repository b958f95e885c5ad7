#!/bin/bash
# copies the summaries of the closing run (tools/final_run.sh -> gpurun_out/) into profiles/ under their tracked names:
#   tools/publish_profiles.sh r05
set -e
R=${1:?round tag, e.g. r05}
S=gpurun_out/profile_summary_vit_b; F=gpurun_out/final
cp $F/bench_vit_b_b1.json profiles/${R}_bench_vit_b_b1.json
cp $F/bench_vit_b_b1_steps200.json profiles/${R}_bench_vit_b_b1_steps200.json
cp $F/bench_vit_h_b1.json profiles/${R}_bench_vit_h_b1.json
cp $F/configs_vit_b.txt profiles/${R}_configs_vit_b.txt
cp $F/configs_vit_h.txt profiles/${R}_configs_vit_h.txt
cp $F/bench_gpus2_rehearsal.json profiles/${R}_bench_gpus2_rehearsal_one_gpu.json
cp $S/hbm_traffic_pmc.json profiles/${R}_hbm_traffic_pmc.json
cp $S/mfma_util_pmc.json profiles/${R}_mfma_util_pmc.json
cp $S/lanes_summary.txt profiles/${R}_lanes_summary.txt
cp $S/kernel_stats_single_lane.txt profiles/${R}_bench_vit_b_b1_kernel_stats_single_lane.txt
cp $S/kernel_stats_single_lane_by_grid.txt profiles/${R}_bench_vit_b_b1_kernel_stats_single_lane_by_grid.txt
cp $S/kernel_stats_rocprofv3.csv profiles/${R}_bench_vit_b_b1_kernel_stats.csv
cp $S/bench_under_kernel_trace.json profiles/${R}_bench_under_kernel_trace.json
cp $S/kernel_stats_burst_process.txt profiles/${R}_kernel_stats_burst_process.txt
cp $F/power_kernels_vit_b.txt profiles/${R}_power_kernels_vit_b.txt
cp $F/defer_probe.txt profiles/${R}_defer_probe.txt
{ echo "# name	value	tolerance (last GPU suite run of the round; every value the parity tests checked)"; cat gpurun_out/parity_margins.txt; } > profiles/${R}_parity_margins.txt
