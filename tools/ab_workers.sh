#!/bin/bash
# same-box A/B of the lane enqueue threads (DLIMGEDIT_STEP_WORKERS=0: the caller enqueues, as before): queue / concurrency
# tests first, then bench.py interleaved, then the enqueue timing of one burst either way
set -e
mkdir -p gpurun_out/abw
timeout -k 10 900 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/abw/tests.log 2>&1 || { tail -30 gpurun_out/abw/tests.log; exit 1; }
tail -2 gpurun_out/abw/tests.log
for round in 1 2 3; do
  DLIMGEDIT_STEP_WORKERS=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abw/inline_$round.json 2> gpurun_out/abw/inline_$round.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abw/workers_$round.json 2> gpurun_out/abw/workers_$round.err
  python - <<PY
import json
for n in ("inline", "workers"):
    d = json.loads(open(f"gpurun_out/abw/{n}_$round.json").read().strip().splitlines()[-1])
    print("$round", n, round(d["value"], 1), d.get("value_min_max"), flush=True)
PY
done
DLIMGEDIT_STEP_WORKERS=0 timeout -k 10 120 python tools/enqueue_time.py 20 9 > gpurun_out/abw/enqueue_inline.txt 2>&1
timeout -k 10 120 python tools/enqueue_time.py 20 9 > gpurun_out/abw/enqueue_workers.txt 2>&1
cat gpurun_out/abw/enqueue_inline.txt gpurun_out/abw/enqueue_workers.txt
timeout -k 10 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --repeats 5 > gpurun_out/abw/workers_steady.json 2> gpurun_out/abw/workers_steady.err
python -c "
import json
d=json.loads(open('gpurun_out/abw/workers_steady.json').read().strip().splitlines()[-1]); print('steady', d['value'])"
timeout -k 10 300 python bench.py --model vit_h --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/abw/vit_h.json 2> gpurun_out/abw/vit_h.err
python -c "
import json
d=json.loads(open('gpurun_out/abw/vit_h.json').read().strip().splitlines()[-1]); print('vit_h', d['value'])"
