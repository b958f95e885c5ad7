#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe3"
mkdir -p "$O"
cd "$R"
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-abi-path"
for C in 2 3 4; do
    DLIMGEDIT_COALESCE=$C $B --steps 20 > "$O/c${C}_s20.json" 2> "$O/c${C}_s20.err" && echo c$C s20 ok
done
DLIMGEDIT_COALESCE=2 $B --steps 200 > "$O/c2_s200.json" 2> "$O/c2_s200.err" && echo ok
DLIMGEDIT_COALESCE=3 $B --steps 200 > "$O/c3_s200.json" 2> "$O/c3_s200.err" && echo ok
DLIMGEDIT_COALESCE=2 DLIMGEDIT_LANES=3 $B --steps 20 > "$O/c2_l3_s20.json" 2> "$O/c2_l3_s20.err" && echo ok
DLIMGEDIT_COALESCE=3 DLIMGEDIT_LANES=3 $B --steps 20 > "$O/c3_l3_s20.json" 2> "$O/c3_l3_s20.err" && echo ok
timeout -k 10 300 python3 -m pytest tests/test_gpu_concurrency.py -x -q -m gpu > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe3")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f" % d["value"], "minmax", [round(v) for v in d["value_min_max"]], "chip %.3f" % r["chip_frac"], "frac %.3f" % r["frac"],
              "alone %.3f" % r.get("frac_single_lane", 0))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
