import sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path("/root/repo")
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config
cfg = get_config("vit_b")
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    view = api.ImageView(synthetic_image(0), api.Channels.rgba)
    for _ in range(8):
        seg = api.Segmentation.process(view, env); seg.compute_mask(api.Point(512, 512))
    def loop(name, fn, n=60):
        c0 = api.ext.queue_config(env)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        c1 = api.ext.queue_config(env)
        ts = np.array(ts) * 1e3
        print(f"{name:44s} median {np.median(ts):.3f} ms  min {ts.min():.3f}  p90 {np.percentile(ts, 90):.3f}   one-image passes {c1['one_image_passes'] - c0['one_image_passes']}, alone {c1['one_image_passes_alone'] - c0['one_image_passes_alone']}")
    keep = []
    loop("process, handle dropped at once", lambda: api.Segmentation.process(view, env))
    loop("process, handles kept", lambda: keep.append(api.Segmentation.process(view, env)))
    keep.clear()
    def both():
        s = api.Segmentation.process(view, env); s.compute_mask(api.Point(512, 512))
    loop("process + compute_mask", both)
    seg = api.Segmentation.process(view, env)
    loop("compute_mask", lambda: seg.compute_mask(api.Point(512, 512)))
