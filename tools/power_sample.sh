#!/bin/bash
# package power and shader clock while bench.py runs a long block (evidence for the power-limit statement in DESIGN.md)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
python3 bench.py --steps 3000 --warmup 5 --repeats 3 --no-cpu-baseline --no-abi-path > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -4
  echo "--"
  sleep 1
done
wait $BP
grep "^{" gpurun_out/power_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'images/s over', d['steps'], 'steps')"
