#!/bin/bash
# lanes x coalescing width x queue depth at the official block length, final code (same box)
out=gpurun_out/r03_probe8; mkdir -p $out
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path > $out/$name.json 2>/dev/null; python3 - $out/$name.json $name <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(j['value'],1))
PY
}
run base A=1
run lanes5 DLIMGEDIT_LANES=5
run lanes3 DLIMGEDIT_LANES=3
run co3 DLIMGEDIT_COALESCE=3
run co4 DLIMGEDIT_COALESCE=4
run co1 DLIMGEDIT_COALESCE=1
run depth1 DLIMGEDIT_STEP_DEPTH=1
run depth3 DLIMGEDIT_STEP_DEPTH=3
run lanes5co1 DLIMGEDIT_LANES=5 DLIMGEDIT_COALESCE=1
run base2 A=1
