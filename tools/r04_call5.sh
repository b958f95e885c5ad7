#!/bin/bash
# round 4, GPU call 5: persistent ping-pong GEMM -- parity, in-kernel stamps, A/B of the bench with chaining on / off
set -o pipefail
O=gpurun_out/r04_c5; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc $?"; tail -5 $O/pytest_gemm.log
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py tests/test_gpu_e2e.py -m gpu -x -q > $O/pytest_e2e.log 2>&1; echo "pytest e2e rc $?"; tail -5 $O/pytest_e2e.log
for p in 1 0; do
  echo "== chaining $p"
  DLIMGEDIT_GEMM_PERSIST=$p DLIMGEDIT_TUNING_LIB=1 timeout -k 10 200 python tools/gemm_clock2.py 8192 2>&1 | tee $O/clock2_persist$p.log
done
for rep in 1 2; do for p in 1 0; do
  DLIMGEDIT_GEMM_PERSIST=$p DLIMGEDIT_TUNING_LIB=1 timeout -k 10 300 python bench.py --steps 20 --no-abi-path --no-cpu-baseline > $O/bench_persist${p}_$rep.json 2> $O/bench_persist${p}_$rep.err; echo "bench persist=$p rc $?"
  python tools/show_bench.py r04_c5/bench_persist${p}_$rep 2>/dev/null | cut -c1-200
done; done
