#!/bin/bash
# dispatch-depth probe: same box, bench at --steps 20 under several DLIMGEDIT_STEP_DEPTH values
set -e
out=gpurun_out/r03_probe7; mkdir -p $out
for rep in 1 2; do
for d in 64 2 1 3; do
  DLIMGEDIT_STEP_DEPTH=$d python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/depth${d}_$rep.json 2> $out/depth${d}_$rep.err
  echo "depth $d rep $rep ok"
done; done
DLIMGEDIT_STEP_DEPTH=2 python3 bench.py --gpus 1 --steps 200 --warmup 5 > $out/depth2_steps200.json 2> $out/depth2_steps200.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_probe7/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['value'], j.get('abi_path',{}).get('one_thread'))
    except Exception as e: print(f, 'ERR', e)
PY
