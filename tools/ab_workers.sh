#!/bin/bash
# same-box A/B of the lane enqueue threads (DLIMGEDIT_STEP_WORKERS=0: the caller enqueues, as before): the whole GPU suite
# first, then bench.py interleaved (official line + the ABI figures)
set -e
mkdir -p gpurun_out/abw
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/abw/tests.log 2>&1 || { tail -30 gpurun_out/abw/tests.log; exit 1; }
tail -2 gpurun_out/abw/tests.log
for round in 1 2; do
  DLIMGEDIT_STEP_WORKERS=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abw/inline_$round.json 2> gpurun_out/abw/inline_$round.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abw/workers_$round.json 2> gpurun_out/abw/workers_$round.err
  python - <<PY
import json
for n in ("inline", "workers"):
    d = json.loads(open(f"gpurun_out/abw/{n}_$round.json").read().strip().splitlines()[-1])
    a = d["abi_path"]
    print("$round", n, round(d["value"], 1), {k: round(v, 1) for k, v in a.items() if isinstance(v, float)}, flush=True)
PY
done
DLIMGEDIT_STEP_WORKERS=0 timeout -k 10 300 python tools/bench_configs.py > gpurun_out/abw/configs_inline.txt 2>&1
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/abw/configs_workers.txt 2>&1
tail -8 gpurun_out/abw/configs_inline.txt; tail -8 gpurun_out/abw/configs_workers.txt
