#!/bin/bash
# profiles of the bench command: kernel traces (all lanes / single lane), the lanes summary, the three counter
# passes (separate --pmc runs: FETCH_SIZE, WRITE_SIZE, MFMA), the profiler's own --stats csv.  Summaries only travel back.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
MODEL="${1:-vit_b}"
export GPU_MAX_HW_QUEUES=8 DLIMGEDIT_PLAIN_STREAMS=1
O=$R/gpurun_out/prof_$MODEL
rm -rf "$O"; mkdir -p "$O"
B="python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-abi-path --no-config-legs --repeats 5 --model $MODEL"
DEPTH=$(python3 -c "import sys; sys.path.insert(0, '$R'); from dlimgedit_amd.sam_config import get_config; print(get_config('$MODEL').depth)")
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/kt -o kt -- $B > $O/bench_kt.log 2>&1 && echo kt ok &&
DLIMGEDIT_SINGLE_LANE=1 timeout -k 10 200 rocprofv3 --kernel-trace -d $O/kt1 -o kt1 -- $B > $O/bench_kt1.log 2>&1 && echo kt1 ok &&
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- $B > $O/f.log 2>&1 && echo fetch ok &&
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write -- $B > $O/w.log 2>&1 && echo write ok &&
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/pmc_mfma -o m -- $B > $O/m.log 2>&1 && echo mfma ok
timeout -k 10 250 rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- $B > $O/st.log 2>&1 && echo stats ok
S=$R/gpurun_out/profile_summary_$MODEL; rm -rf "$S"; mkdir -p "$S"
FLOP=$(python3 -c "import sys; sys.path.insert(0, '$R'); from dlimgedit_amd.sam_config import get_config; print(get_config('$MODEL').encoder_flops() + 3.62e9)")
# the same trace of the timed region alone (bursts of 20 requests + synchronize): block structure is unambiguous there
DLIMGEDIT_SAM_MODEL=$MODEL timeout -k 10 250 rocprofv3 --kernel-trace -d $O/ktb -o ktb -- python3 $R/tools/burst_trace.py 20 8 > $O/burst.log 2>&1 && echo burst ok
{ echo "# rocprofv3 --kernel-trace of tools/burst_trace.py 20 8 ($MODEL): the timed region of bench.py (20 requests + synchronize)."
  echo "# Whether the lanes overlap under the profiler shows in the per-lane pass starts below and in the burst rate against"
  echo "# the unprofiled 'value'; the overlapped regime's own clocks are in the bench line (roofline.under_lanes, HIP events)."
  grep "images/s" $O/burst.log
  python3 $R/tools/trace_lanes.py $O/ktb/ktb_results.db 2>&1 | tail -5
  python3 $R/tools/lanes_summary.py $O/ktb/ktb_results.db 20 $FLOP; } > $S/lanes_summary.txt 2>&1
# the burst process as a whole (model load + 8 bursts of 20 requests, no hbm_kernels leg, no ABI legs): its only blit copies
# (__amd_rocclr_copyBuffer) are the ~170 weight tensors of the model load; the per-block table of lanes_summary.txt -- the
# timed region proper -- has none.  The one-workgroup copies of the bench trace are the set-up of the hbm_kernels timing loop
# (16-byte IoU vectors, 256 KB logit planes of its 768 MB ring) and the model load.
python3 $R/tools/kernel_stats.py $O/ktb/ktb_results.db 40 > $S/kernel_stats_burst_process.txt 2>&1
python3 $R/tools/kernel_stats.py $O/kt1/kt1_results.db 40 > $S/kernel_stats_single_lane.txt
python3 $R/tools/kernel_stats.py $O/kt1/kt1_results.db 60 --by-grid > $S/kernel_stats_single_lane_by_grid.txt 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/fetch_results.db $O/pmc_write/write_results.db $S/hbm_traffic_pmc.json "$B" $DEPTH > $S/traffic.log 2>&1
python3 $R/tools/pmc_mfma.py $O/pmc_mfma/m_results.db $S/mfma_util_pmc.json "$B" > $S/mfma.log 2>&1
cp $(ls $O/st/*/st_kernel_stats.csv $O/st/st_kernel_stats.csv 2>/dev/null | head -1) $S/kernel_stats_rocprofv3.csv 2>/dev/null
grep "^{" $O/bench_kt.log | tail -1 > $S/bench_under_kernel_trace.json
rm -rf "$O"
ls -la $S
