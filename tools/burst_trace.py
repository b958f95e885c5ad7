"""Runs only bench.py's timed region (bursts of K single-image requests + synchronize) with a pause between bursts, so that a
rocprofv3 --kernel-trace shows each burst as one block:   rocprofv3 --kernel-trace ... -- python3 tools/burst_trace.py [K] [bursts]
then tools/trace_lanes.py / tools/lanes_summary.py on the .db."""
import os
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from bench import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bursts = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pause = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0005      # seconds between bursts (a long pause lets the clocks drop)
model = os.environ.get("DLIMGEDIT_SAM_MODEL", "vit_b")
cfg = get_config(model)
model_dir = os.path.join(tempfile.gettempdir(), f"dlimgedit_bench_{model}_7_{os.getuid()}")
target = Path(model_dir) / "segmentation" / W.weight_file_name(cfg)
if not target.exists():
    W.save_weights(target, cfg, W.synthetic_weights(cfg, 7))
os.environ["DLIMGEDIT_SAM_MODEL"] = model
env = api.Environment(api.Options(api.Backend.gpu, model_dir))
ext = api.ext
im = synthetic_image(0)
p = ext.device_alloc(env, im.nbytes)
ext.copy_to_device(env, p, im)
mask = ext.device_alloc(env, 1024 * 1024)
views = ext.device_views([p], 1024, 1024)
points = [api.Point(512, 512)]
for _ in range(5):
    ext.encode_and_mask(env, views, points, [mask])
ext.synchronize(env)
times = []
for _ in range(bursts):
    time.sleep(pause)
    t0 = time.perf_counter()
    for _ in range(K):
        ext.encode_and_mask(env, views, points, [mask])
    ext.synchronize(env)
    times.append(time.perf_counter() - t0)
print("burst ms:", " ".join(f"{1e3 * t:.2f}" for t in times), "| images/s at the median:", f"{K / float(np.median(times)):.1f}")
