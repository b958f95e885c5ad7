#!/bin/bash
# same-box A/B of the single mask written straight to pinned host memory (DLIMGEDIT_DIRECT_MASKS=0: device buffer + copy)
mkdir -p gpurun_out/abd
for round in 1 2; do
  for d in 0 1; do
    DLIMGEDIT_DIRECT_MASKS=$d timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abd/d${d}_$round.json 2> gpurun_out/abd/d${d}_$round.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/abd/d${d}_$round.json").read().strip().splitlines()[-1])
print("$round direct=$d", round(d["value"], 1), {k: round(v) for k, v in d["decode_only"].items() if isinstance(v, float)}, {k: round(v) for k, v in d["abi_path"].items() if isinstance(v, float)}, flush=True)
PY
  done
done
