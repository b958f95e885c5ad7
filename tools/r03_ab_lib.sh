#!/bin/bash
# same-box A/B of the product library against a copy of a tuning build (lib/libdlimgedit_<name>.so): bench value, alternating
name=$1; out=gpurun_out/r03_ab; mkdir -p $out
for i in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path > $out/prod_$i.json 2>/dev/null
  DLIMGEDIT_TUNING_LIB=libdlimgedit_$name.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path > $out/${name}_$i.json 2>/dev/null
done
python3 - $out $name <<'PY'
import json,sys,glob
for tag in ("prod", sys.argv[2]):
    vals=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"{sys.argv[1]}/{tag}_*.json"))]
    print(tag, [round(v['value'],1) for v in vals], {k: round(v['ms_per_step'],3) for k,v in vals[-1]['stages'].items() if isinstance(v,dict) and 'ms_per_step' in v})
PY
