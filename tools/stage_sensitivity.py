"""How much of the whole-job rate each part of the path costs when the lanes overlap: images/s of encode+mask vs
encode only (same process, same box).  usage: python tools/stage_sensitivity.py [steps]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
from conftest import synthetic_image  # noqa: E402
from dlimgedit_amd import api, weights as W  # noqa: E402
from dlimgedit_amd.sam_config import CONFIGS  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    model = os.environ.get("DLIMGEDIT_SAM_MODEL", "vit_b")
    cfg = CONFIGS[model]
    mdir = Path(os.environ.get("TMPDIR", "/tmp")) / "dlimgedit_models"
    target = mdir / "segmentation" / f"sam_{model}.dlw"
    if not target.exists():
        target.parent.mkdir(parents=True, exist_ok=True)
        W.save_weights(target, cfg, W.synthetic_weights(cfg, 0))
    os.environ["DLIMGEDIT_SAM_MODEL"] = model
    env = api.Environment(api.Options(api.Backend.gpu, str(mdir)))
    ext = api.ext
    img = synthetic_image(0)
    p = ext.device_alloc(env, img.nbytes)
    ext.copy_to_device(env, p, img)
    mask = ext.device_alloc(env, 1024 * 1024)
    views = ext.device_views([p], 1024, 1024)
    pts = [api.Point(512, 512)]

    def run(fn):
        for _ in range(5):
            fn()
        ext.synchronize(env)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        t_enqueue = time.perf_counter() - t0
        ext.synchronize(env)
        run.enqueue_ms = 1e3 * t_enqueue / steps          # host time to enqueue one step
        return steps / (time.perf_counter() - t0)

    for rep in range(2):
        full = run(lambda: ext.encode_and_mask(env, views, pts, [mask]))
        enc = run(lambda: ext.encode_only(env, views))
        print(f"lanes {ext.lane_count(env)}: encode+mask {full:7.1f} images/s   encode only {enc:7.1f} images/s   "
              f"(decoder+post cost {1e3 / full - 1e3 / enc:5.3f} ms/image; host enqueue {run.enqueue_ms:5.3f} ms/step)", flush=True)


if __name__ == "__main__":
    main()
