#!/bin/bash
# step queue of the larger models (MODEL=vit_h default, vit_l): requests per pass x lanes the step queue deals to, one box, two rounds interleaved; STEPS (default 20) per block
#   [MODEL=vit_l] [STEPS=50] tools/sweep_queue.sh "2 0" "3 2" ...        (pairs "coalesce step_lanes"; step_lanes 0 = every lane)
STEPS=${STEPS:-20}   # MODEL=vit_l sweeps another model
mkdir -p gpurun_out/sweep_h
[ $# -gt 0 ] || set -- "2 0" "3 0" "4 0" "2 2" "3 2" "4 2" "4 3"
for round in 1 2; do
  for cfg in "$@"; do
    c=${cfg% *}; l=${cfg#* }
    out=gpurun_out/sweep_h/s${STEPS}_c${c}_l${l}_$round
    DLIMGEDIT_COALESCE=$c DLIMGEDIT_STEP_LANES=$l timeout -k 10 300 python3 bench.py --model ${MODEL:-vit_h} --steps $STEPS --warmup 4 --repeats 9 --no-abi-path --no-cpu-baseline --no-config-legs > $out.json 2> $out.err || { echo "coalesce $c lanes $l FAILED"; tail -3 $out.err; continue; }
    python3 - "$out.json" "$round" "$c" "$l" "$STEPS" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("steps", sys.argv[5], "round", sys.argv[2], "coalesce", sys.argv[3], "step lanes", sys.argv[4], ":", round(d["value"], 1), "images/s", flush=True)
PY
  done
done
