"""One-line summaries of bench.py outputs: python tools/show_bench.py NAME ... reads gpurun_out/NAME.log or .json"""
import json
import sys
from pathlib import Path

for f in sys.argv[1:]:
    p = next((q for q in (Path("gpurun_out") / (f + ext) for ext in (".log", ".json", "")) if q.exists()), None)
    d = json.loads(p.read_text().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value", round(d["value"], 1), d.get("value_min_max") and [round(v) for v in d["value_min_max"]],
          "| gemm", round(r["achieved"], 1), "TF frac", round(r["frac"], 3), "chip", round(r.get("chip_frac", 0), 3),
          "| iou", round(d.get("mask_iou", -1), 5), {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()},
          "| abi", {k: round(v) for k, v in d.get("abi_path", {}).items() if isinstance(v, float)})
