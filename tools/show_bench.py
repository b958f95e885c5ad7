"""One-line summaries of bench.py outputs: python tools/show_bench.py NAME ... reads gpurun_out/NAME.log or .json"""
import json
import sys
from pathlib import Path

for f in sys.argv[1:]:
    p = next((q for q in (Path("gpurun_out") / (f + ext) for ext in (".log", ".json", "")) if q.exists()), None)
    d = json.loads(p.read_text().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["roofline"]["achieved"], 1), round(d["mask_iou"], 5),
          {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
