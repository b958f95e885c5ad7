#!/bin/bash
# round-3 closing run: the GPU tests, the default bench line, the profile collection -- everything profiles/ and DESIGN quote
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_final"
mkdir -p "$O"
cd "$R"
rm -f gpurun_out/parity_margins.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
cp gpurun_out/parity_margins.txt "$O/parity_margins.txt" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > "$O/bench_vit_b_b1.json" 2> "$O/bench_vit_b_b1.err"; echo "bench rc=$?"
python3 bench.py --steps 200 --warmup 5 --repeats 7 --no-cpu-baseline --no-abi-path > "$O/bench_vit_b_b1_steps200.json" 2>/dev/null
bash tools/collect_profiles_r03.sh > "$O/collect.log" 2>&1; tail -3 "$O/collect.log"
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_final")
for f in sorted(glob.glob(O + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(os.path.basename(f), "value %.1f" % d["value"], "chip %.3f" % r["chip_frac"], "frac %.3f" % r["frac"], "alone %.3f" % r["frac_single_lane"], d.get("hbm_kernels", {}).get("batch16"))
PY
