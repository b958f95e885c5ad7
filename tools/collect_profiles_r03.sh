#!/bin/bash
# round-3 profiles: kernel traces (4 lanes / single lane) and the three counter passes of the bench command
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export GPU_MAX_HW_QUEUES=8 DLIMGEDIT_PLAIN_STREAMS=1
O=$R/gpurun_out/r03f
rm -rf "$O"; mkdir -p "$O"
B="python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-abi-path --repeats 5"
timeout -k 10 150 rocprofv3 --kernel-trace -d $O/kt -o kt -- $B > $O/bench_kt.log 2>&1 && echo kt ok &&
DLIMGEDIT_SINGLE_LANE=1 timeout -k 10 150 rocprofv3 --kernel-trace -d $O/kt1 -o kt1 -- $B > $O/bench_kt1.log 2>&1 && echo kt1 ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch -- $B > $O/f.log 2>&1 && echo fetch ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write -- $B > $O/w.log 2>&1 && echo write ok &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/pmc_mfma -o m -- $B > $O/m.log 2>&1 && echo mfma ok

# the profiler's own --stats table of the 4-lane run (csv)
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/st -o st --output-format csv -- $B > $O/st.log 2>&1 && echo stats ok
# summaries only travel back (the databases are larger than gpurun's 64 MiB return limit)
S=$R/gpurun_out/r03_summary; rm -rf "$S"; mkdir -p "$S"
python3 $R/tools/kernel_stats.py $O/kt/kt_results.db 40 > $S/kernel_stats_4lanes.txt
python3 $R/tools/kernel_stats.py $O/kt1/kt1_results.db 40 > $S/kernel_stats_single_lane.txt
python3 $R/tools/kernel_stats.py $O/kt1/kt1_results.db 60 --by-grid > $S/kernel_stats_single_lane_by_grid.txt 2>&1
python3 - "$O/kt1/kt1_results.db" > $S/dispatch_columns.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
print(kd, [r[1] for r in db.execute(f"pragma table_info({kd})")])
PY
python3 $R/tools/pmc_traffic.py $O/pmc_fetch/fetch_results.db $O/pmc_write/write_results.db $S/hbm_traffic_pmc.json "$B" > $S/traffic.log 2>&1
python3 $R/tools/pmc_mfma.py $O/pmc_mfma/m_results.db $S/mfma_util_pmc.json "$B" > $S/mfma.log 2>&1
cp $(ls $O/st/*/st_kernel_stats.csv $O/st/st_kernel_stats.csv 2>/dev/null | head -1) $S/kernel_stats_rocprofv3.csv 2>/dev/null
grep "^{" $O/bench_kt.log | tail -1 > $S/bench_under_kernel_trace.json
rm -rf "$O"
ls -la $S
