"""Host time of the enqueue alone: K encode_and_mask requests issued back to back (no synchronisation in between), the
time until the last call returns against the time until the GPU has finished -- how much of a block the host spends
feeding the lanes, i.e. how late the last lane gets its first pass.   python tools/enqueue_time.py [K] [repeats]"""
import os
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
from bench import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
model = os.environ.get("DLIMGEDIT_SAM_MODEL", "vit_b")
cfg = get_config(model)
model_dir = os.path.join(tempfile.gettempdir(), f"dlimgedit_bench_{model}_0_{os.getuid()}")
target = Path(model_dir) / "segmentation" / W.weight_file_name(cfg)
if not target.exists():
    W.save_weights(target, cfg, W.synthetic_weights(cfg, 0))
os.environ["DLIMGEDIT_SAM_MODEL"] = model
env = api.Environment(api.Options(api.Backend.gpu, model_dir))
ext = api.ext
im = synthetic_image(0)
p = ext.device_alloc(env, im.nbytes)
ext.copy_to_device(env, p, im)
mask = ext.device_alloc(env, 1024 * 1024)
views = ext.device_views([p], 1024, 1024)
pts = [api.Point(512, 512)]
for _ in range(6):
    ext.encode_and_mask(env, views, pts, [mask])
ext.synchronize(env)
rows = []
for _ in range(reps):
    t0 = time.perf_counter()
    marks = []
    for _ in range(K):
        ext.encode_and_mask(env, views, pts, [mask])
        marks.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t0
    ext.synchronize(env)
    t_all = time.perf_counter() - t0
    rows.append((t_enq, t_all, marks))
rows.sort(key=lambda r: r[1])
t_enq, t_all, marks = rows[len(rows) // 2]
print(f"{K} requests: last call returned after {t_enq * 1e3:.2f} ms, GPU finished after {t_all * 1e3:.2f} ms ({K / t_all:.0f} images/s)")
print("call return times (ms): " + " ".join(f"{m * 1e3:.2f}" for m in marks))
