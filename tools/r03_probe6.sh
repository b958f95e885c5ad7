#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe6"
mkdir -p "$O"
cd "$R"
rm -f gpurun_out/parity_margins.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
cp gpurun_out/parity_margins.txt "$O/parity_margins.txt" 2>/dev/null
python3 tools/bench_configs.py vit_b 3 > "$O/configs_vit_b.txt" 2>&1; cat "$O/configs_vit_b.txt"
python3 tools/bench_configs.py vit_h 3 > "$O/configs_vit_h.txt" 2>&1; cat "$O/configs_vit_h.txt"
python3 bench.py --model vit_h --steps 10 --warmup 3 --repeats 7 --no-cpu-baseline --no-abi-path > "$O/bench_vit_h.json" 2> "$O/bench_vit_h.err"
python3 bench.py --batch 8 --steps 10 --warmup 3 --repeats 7 --no-cpu-baseline --no-abi-path > "$O/bench_vit_b_b8.json" 2> "$O/bench_vit_b_b8.err"
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe6")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f" % d["value"], "chip %.3f" % r["chip_frac"], "frac %.3f" % r["frac"], "alone %.3f" % r["frac_single_lane"],
              {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
