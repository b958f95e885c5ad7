#!/bin/bash
# same-box A/B of this build against a copy of the previous one (lib/libdlimgedit_head.so): GEMM kernel tests first, then
# bench.py interleaved (official block; steady state at the end)
set -e
mkdir -p gpurun_out/abh
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu > gpurun_out/abh/tests.log 2>&1 || { tail -30 gpurun_out/abh/tests.log; exit 1; }
tail -2 gpurun_out/abh/tests.log
for round in 1 2 3; do
  for n in head new; do
    if [ $n = head ]; then export DLIMGEDIT_TUNING_LIB=libdlimgedit_head.so; else unset DLIMGEDIT_TUNING_LIB; fi
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path > gpurun_out/abh/${n}_$round.json 2> gpurun_out/abh/${n}_$round.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/abh/${n}_$round.json").read().strip().splitlines()[-1])
u = d["roofline"]["under_lanes"]
pk = d["roofline"]["per_kernel"]
print("$round", "$n", round(d["value"], 1), "alone us:", {k: round(v["avg_launch_us"], 1) for k, v in pk.items()}, "gemm under lanes us", round(u["avg_gemm_launch_us"], 1), flush=True)
PY
  done
done
for n in head new; do
  if [ $n = head ]; then export DLIMGEDIT_TUNING_LIB=libdlimgedit_head.so; else unset DLIMGEDIT_TUNING_LIB; fi
  timeout -k 10 300 python bench.py --steps 200 --warmup 5 --repeats 5 --no-cpu-baseline --no-abi-path > gpurun_out/abh/${n}_steady.json 2> gpurun_out/abh/${n}_steady.err
  python -c "
import json
d=json.loads(open('gpurun_out/abh/${n}_steady.json').read().strip().splitlines()[-1]); print('steady $n', round(d['value'],1))"
  timeout -k 10 300 python bench.py --model vit_h --steps 12 --warmup 3 --no-cpu-baseline --no-abi-path > gpurun_out/abh/${n}_vit_h.json 2> gpurun_out/abh/${n}_vit_h.err
  python -c "
import json
d=json.loads(open('gpurun_out/abh/${n}_vit_h.json').read().strip().splitlines()[-1]); print('vit_h $n', round(d['value'],1))"
done
