#!/bin/bash
# same-box A/B of the residual stream representations (one gpurun call): the whole GPU suite, the operand-select probe with
# the v_fma_mix_f32 forms of the pair arithmetic, then bench.py interleaved: previous build (lib/libdlimgedit_head.so),
# this build with the fp32 stream (DLIMGEDIT_SPLIT_STREAM=0), this build with the f16 pair
set -e
mkdir -p gpurun_out/ab
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/ab/suite.log 2>&1 || { tail -30 gpurun_out/ab/suite.log; exit 1; }
tail -2 gpurun_out/ab/suite.log
hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -o /tmp/pkfma_hazard tools/pkfma_hazard.cpp
timeout -k 10 200 /tmp/pkfma_hazard 4 selects > gpurun_out/ab/hazard.txt 2>&1
grep -E "^case (8|12|17|18) " gpurun_out/ab/hazard.txt
for round in 1 2 3; do
  DLIMGEDIT_TUNING_LIB=libdlimgedit_head.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/head_$round.json 2> gpurun_out/ab/head_$round.err
  DLIMGEDIT_SPLIT_STREAM=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/fp32_$round.json 2> gpurun_out/ab/fp32_$round.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/pair_$round.json 2> gpurun_out/ab/pair_$round.err
  python - <<PY
import json
for n in ("head", "fp32", "pair"):
    d = json.loads(open(f"gpurun_out/ab/{n}_$round.json").read().strip().splitlines()[-1])
    print("$round", n, round(d["value"], 1), round(d["roofline"]["frac"], 4), round(d["roofline"]["avg_launch_us"], 2), flush=True)
PY
done
