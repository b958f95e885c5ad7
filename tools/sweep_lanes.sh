#!/bin/bash
# lanes x requests per pass with the lanes' enqueue threads in place (one box): official block (--steps 20) per combination
mkdir -p gpurun_out/sweep
for model in vit_b vit_h; do
  if [ $model = vit_b ]; then LANES="3 4 5 6"; STEPS=20; else LANES="2 3 4"; STEPS=12; fi
  for lanes in $LANES; do
    for co in 1 2 3; do
      DLIMGEDIT_LANES=$lanes DLIMGEDIT_COALESCE=$co timeout -k 10 200 python bench.py --model $model --steps $STEPS --warmup 4 --repeats 11 --no-abi-path --no-cpu-baseline > gpurun_out/sweep/${model}_l${lanes}_c${co}.json 2> gpurun_out/sweep/${model}_l${lanes}_c${co}.err || { echo "$model lanes $lanes coalesce $co FAILED"; tail -3 gpurun_out/sweep/${model}_l${lanes}_c${co}.err; continue; }
      python - <<PY
import json
d = json.loads(open("gpurun_out/sweep/${model}_l${lanes}_c${co}.json").read().strip().splitlines()[-1])
print("$model lanes $lanes coalesce $co:", round(d["value"], 1), d["config"].get("lanes"), d["config"].get("requests_coalesced_per_pass"), flush=True)
PY
    done
  done
done
