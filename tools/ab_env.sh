#!/bin/bash
# Same-box A/B of bench.py under two (or more) environments, interleaved over ROUNDS rounds; one gpurun call.
#   tools/ab_env.sh [-r ROUNDS] [-b "bench args"] "NAME=a" "NAME=b" ...        ("" = the plain environment)
# e.g. the round-4 experiments:
#   tools/ab_env.sh "DLIMGEDIT_SPLIT_STREAM=0" ""                      residual stream fp32 + copy / f16 pair
#   tools/ab_env.sh "DLIMGEDIT_STEP_WORKERS=0" ""                      caller enqueues / the lanes' enqueue threads
#   tools/ab_env.sh "DLIMGEDIT_DIRECT_MASKS=0" ""                      single mask through a copy command / straight to host
#   tools/ab_env.sh "DLIMGEDIT_TUNING_LIB=libdlimgedit_head.so" ""     a copy of another build (lib/libdlimgedit_head.so) / this one
# Per run: images/s, every GEMM flavour's time alone on the chip and under the lanes (a change can hide in one and not in
# the other), the decode and ABI figures when bench.py measured them.
ROUNDS=3; BENCH="--steps 20 --warmup 5 --no-cpu-baseline --no-config-legs"
while getopts "r:b:" o; do case $o in r) ROUNDS=$OPTARG;; b) BENCH=$OPTARG;; esac; done
shift $((OPTIND - 1))
mkdir -p gpurun_out/ab_env
for round in $(seq 1 $ROUNDS); do
  i=0
  for setting in "$@"; do
    i=$((i + 1))
    out=gpurun_out/ab_env/${i}_$round
    if [ -n "$setting" ]; then env $setting timeout -k 10 400 python bench.py $BENCH > $out.json 2> $out.err
    else timeout -k 10 400 python bench.py $BENCH > $out.json 2> $out.err; fi
    python - "$out.json" "$round" "${setting:-(plain)}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
alone = {k: round(v["avg_launch_us"], 1) for k, v in r.get("per_kernel", {}).items()}
lanes = {k: round(v["under_lanes_avg_launch_us"], 1) for k, v in r.get("per_kernel", {}).items()}
extra = {}
for key in ("decode_only", "abi_path"):
    if isinstance(d.get(key), dict):
        extra.update({k: round(v) for k, v in d[key].items() if isinstance(v, float)})
print(sys.argv[2], sys.argv[3], "images/s", round(d["value"], 1), "| alone us", alone, "| under lanes us", lanes, "|", extra, flush=True)
PY
  done
done
