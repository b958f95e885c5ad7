#!/bin/bash
# copies the summaries of the closing run (tools/r04_final.sh -> gpurun_out/) into profiles/ under their tracked names
set -e
S=gpurun_out/r04_summary_vit_b; F=gpurun_out/r04_final
cp $F/bench_vit_b_b1.json profiles/r04_bench_vit_b_b1.json
cp $F/bench_vit_b_b1_steps200.json profiles/r04_bench_vit_b_b1_steps200.json
cp $F/bench_vit_b_b8.json profiles/r04_bench_vit_b_b8.json
cp $F/bench_vit_h_b1.json profiles/r04_bench_vit_h_b1.json
cp $F/configs_vit_b.txt profiles/r04_configs_vit_b.txt
cp $F/configs_vit_h.txt profiles/r04_configs_vit_h.txt
cp $F/pkfma_hazard.txt profiles/r04_pkfma_hazard.txt
cp $S/hbm_traffic_pmc.json profiles/r04_hbm_traffic_pmc.json
cp $S/mfma_util_pmc.json profiles/r04_mfma_util_pmc.json
cp $S/lanes_summary.txt profiles/r04_lanes_summary.txt
cp $S/kernel_stats_single_lane.txt profiles/r04_bench_vit_b_b1_kernel_stats_single_lane.txt
cp $S/kernel_stats_single_lane_by_grid.txt profiles/r04_bench_vit_b_b1_kernel_stats_single_lane_by_grid.txt
cp $S/kernel_stats_rocprofv3.csv profiles/r04_bench_vit_b_b1_kernel_stats.csv
cp $S/bench_under_kernel_trace.json profiles/r04_bench_under_kernel_trace.json
{ echo "# name	value	tolerance (last GPU suite run of round 4; every value the parity tests checked)"; cat gpurun_out/parity_margins.txt; } > profiles/r04_parity_margins.txt
