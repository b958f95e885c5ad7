#!/bin/bash
# round 4, GPU call 6: persistent ping-pong GEMM after the register work -- parity, GEMM table new vs old library, bench A/B
set -o pipefail
O=gpurun_out/r04_c6; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc $?"; tail -3 $O/pytest_gemm.log
cat > /tmp/gt.py <<'PY'
import sys
sys.path.insert(0, ".")
from dlimgedit_amd import api
for name, M, N, K, act, fl in [("qkv", 8192, 2304, 768, 0, 1), ("fc1", 8192, 3072, 768, 1, 1), ("proj", 8192, 768, 768, 0, 3), ("fc2", 8192, 768, 3072, 0, 3),
                               ("qkv1", 4096, 2304, 768, 0, 1), ("fc1_1", 4096, 3072, 768, 1, 1), ("qkv8", 32768, 2304, 768, 0, 1), ("fc1_8", 32768, 3072, 768, 1, 1)]:
    m1 = api.ext.bench_gemm(M, N, K, act, iters=40, flavour=fl, tile=9, shared=True, streams=1)
    m4 = api.ext.bench_gemm(M, N, K, act, iters=20, flavour=fl, tile=9, shared=True, streams=4)
    gf = 2.0 * M * N * K / 1e9
    print(f"  {name:6s} M={M:5d} 1 stream {m1 * 1e3:7.1f} us {gf / m1:6.0f} TF | 4 streams {m4 * 1e3:7.1f} us/GEMM {gf / m4:6.0f} TF", flush=True)
PY
for lib in new old; do
  echo "== library $lib"
  if [ $lib = old ]; then export DLIMGEDIT_TUNING_LIB=libdlimgedit_old.so; else unset DLIMGEDIT_TUNING_LIB; fi
  timeout -k 10 200 python /tmp/gt.py 2>&1 | tee $O/table_$lib.log
done
for rep in 1 2; do for lib in new old; do
  if [ $lib = old ]; then export DLIMGEDIT_TUNING_LIB=libdlimgedit_old.so; else unset DLIMGEDIT_TUNING_LIB; fi
  timeout -k 10 300 python bench.py --steps 20 --no-abi-path --no-cpu-baseline > $O/bench_${lib}_$rep.json 2> $O/bench_${lib}_$rep.err; echo "bench $lib rc $?"
  python tools/show_bench.py r04_c6/bench_${lib}_$rep 2>/dev/null | cut -c1-220
done; done
unset DLIMGEDIT_TUNING_LIB
echo "== batch 8"
for lib in new old; do
  if [ $lib = old ]; then export DLIMGEDIT_TUNING_LIB=libdlimgedit_old.so; else unset DLIMGEDIT_TUNING_LIB; fi
  timeout -k 10 300 python bench.py --steps 10 --batch 8 --no-abi-path --no-cpu-baseline > $O/bench8_${lib}.json 2> $O/bench8_${lib}.err; echo "bench8 $lib rc $?"
  python tools/show_bench.py r04_c6/bench8_${lib} 2>/dev/null | cut -c1-220
done
