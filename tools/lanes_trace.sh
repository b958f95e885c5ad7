#!/bin/bash
# kernel trace of the bench command with all lanes + tools/lanes_summary.py on it (the database stays on the GPU box)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:?run through gpurun}"
MODEL="${1:-vit_b}"; STEPS="${2:-20}"
export GPU_MAX_HW_QUEUES=8 DLIMGEDIT_PLAIN_STREAMS=1
O=$R/gpurun_out/lanes_$MODEL; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --steps $STEPS --warmup 4 --no-cpu-baseline --no-abi-path --repeats 7 --model $MODEL"
timeout -k 10 250 rocprofv3 --kernel-trace -d $O/kt -o kt -- $B > $O/bench_kt.log 2>&1 && echo kt ok
FLOP=$(python3 -c "import sys; sys.path.insert(0, '$R'); from dlimgedit_amd.sam_config import get_config; print(get_config('$MODEL').encoder_flops() + 3.62e9)")
python3 $R/tools/lanes_summary.py $O/kt/kt_results.db $STEPS $FLOP --blocks > $R/gpurun_out/lanes_summary_$MODEL.txt 2>&1
grep "^{" $O/bench_kt.log | tail -1 > $R/gpurun_out/lanes_bench_$MODEL.json
rm -rf $O
