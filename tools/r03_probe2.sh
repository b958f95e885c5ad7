#!/bin/bash
# round-3: GPU tests after the host-side changes, then coalescing / tile sweeps of the bench and the 2-rank rehearsal
set -o pipefail
R="${GRAFT_REPO_ROOT:?}"
O="$R/gpurun_out/r03_probe2"
mkdir -p "$O"
cd "$R"
rm -f gpurun_out/parity_margins.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$O/pytest.log"
cp gpurun_out/parity_margins.txt "$O/parity_margins.txt" 2>/dev/null
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-abi-path"
for C in 1 2 4; do
    DLIMGEDIT_COALESCE=$C $B > "$O/coalesce$C.json" 2> "$O/coalesce$C.err" && echo coalesce $C ok
done
DLIMGEDIT_COALESCE=2 DLIMGEDIT_GEMM_BATCH_PP=0 $B > "$O/coalesce2_pp128.json" 2> "$O/coalesce2_pp128.err" && echo coalesce 2 pp128 ok
DLIMGEDIT_COALESCE=2 DLIMGEDIT_LANES=3 $B > "$O/coalesce2_l3.json" 2> "$O/coalesce2_l3.err" && echo coalesce 2 lanes 3 ok
DLIMGEDIT_COALESCE=4 DLIMGEDIT_LANES=2 $B > "$O/coalesce4_l2.json" 2> "$O/coalesce4_l2.err" && echo coalesce 4 lanes 2 ok
DLIMGEDIT_COALESCE=2 $B --steps 200 > "$O/coalesce2_s200.json" 2> "$O/coalesce2_s200.err" && echo steps200 ok
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 10 --warmup 3 --repeats 5 --rehearse-gloo --no-cpu-baseline --no-abi-path > "$O/rehearse2.json" 2> "$O/rehearse2.err"; echo "rehearse rc=$?"
python3 - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_probe2")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), "value %.1f" % d["value"], "chip %.3f" % r["chip_frac"], "frac %.3f" % r["frac"],
              "alone %.3f" % r.get("frac_single_lane", 0), {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()}, d.get("rccl"))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
tail -3 "$O/rehearse2.err"
