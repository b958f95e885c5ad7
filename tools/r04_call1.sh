#!/bin/bash
# round 4, GPU call 1: suite on the hygiene build, the packed-FMA hazard probe, baseline bench line, GEMM stamps (two-image shapes)
set -o pipefail
O=gpurun_out/r04_c1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log
timeout -k 10 120 tools/_bin/pkfma_hazard 4 > $O/pkfma.log 2>&1; echo "pkfma rc $?"; cat $O/pkfma.log
timeout -k 10 300 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python tools/show_bench.py r04_c1/bench 2>/dev/null | head -20
DLIMGEDIT_TUNING_LIB=1 timeout -k 10 200 python tools/gemm_clock2.py 8192 > $O/clock2.log 2>&1; cat $O/clock2.log
DLIMGEDIT_TUNING_LIB=1 timeout -k 10 200 python tools/gemm_clock2.py 4096 > $O/clock1.log 2>&1; cat $O/clock1.log
