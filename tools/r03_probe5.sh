#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe5"
mkdir -p "$O"
cd "$R"
python3 tools/gemm_table.py vit_b -1 > "$O/gemm_table.txt" 2>&1; grep -E "auto/shared|GF" "$O/gemm_table.txt"
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-abi-path --repeats 11"
for L in 2 3 4; do for C in 2 3 4 5; do
    DLIMGEDIT_LANES=$L DLIMGEDIT_COALESCE=$C $B --steps 20 > "$O/l${L}_c${C}_s20.json" 2> "$O/l${L}_c${C}_s20.err"
    DLIMGEDIT_LANES=$L DLIMGEDIT_COALESCE=$C $B --steps 120 > "$O/l${L}_c${C}_s120.json" 2> "$O/l${L}_c${C}_s120.err"
done; done
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe5")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.1f" % d["value"], "max %.0f" % d["value_min_max"][1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
