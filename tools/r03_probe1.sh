#!/bin/bash
# round-3 first look: the bench as it stands, lanes and batch sweeps, hipBLASLt beside our GEMMs under 4 streams
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe1"
mkdir -p "$O"
cd "$R"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path"
$B > "$O/base.json" 2> "$O/base.err" && echo base ok
for L in 3 5 6 8; do
    DLIMGEDIT_LANES=$L $B > "$O/lanes$L.json" 2> "$O/lanes$L.err" && echo lanes $L ok
done
for BT in 2 4; do
    $B --batch $BT > "$O/batch$BT.json" 2> "$O/batch$BT.err" && echo batch $BT ok
    DLIMGEDIT_LANES=2 $B --batch $BT > "$O/batch${BT}_l2.json" 2> "$O/batch${BT}_l2.err" && echo batch $BT lanes 2 ok
done
$B --steps 200 > "$O/steps200.json" 2> "$O/steps200.err" && echo steps200 ok
python3 tools/power_gemm.py 3 > "$O/power_gemm.txt" 2>&1 && echo power ok
python3 - <<'EOF'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe1")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f" % d["value"], "chip %.3f" % r["chip_frac"], "frac %.3f" % r["frac"],
              {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
EOF
cat "$O/power_gemm.txt"
