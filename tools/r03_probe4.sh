#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe4"
mkdir -p "$O"
cd "$R"
rm -f gpurun_out/parity_margins.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_concurrency.py -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
grep -E "golden|e2e.embedding|e2e.logits" gpurun_out/parity_margins.txt | head -20
B="python3 bench.py --warmup 5 --no-cpu-baseline --no-abi-path"
$B --steps 20 > "$O/s20.json" 2> "$O/s20.err" && echo ok
$B --steps 200 > "$O/s200.json" 2> "$O/s200.err" && echo ok
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe4")
for f in sorted(glob.glob(O + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(os.path.basename(f), "value %.1f" % d["value"], "minmax", [round(v) for v in d["value_min_max"]], "chip %.3f" % r["chip_frac"],
          {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
PY
