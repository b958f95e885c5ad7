"""Where the time of the drop-in ABI path goes: single calls and the batch slots, one host thread, wall clock.
python tools/abi_phases.py [vit_b|vit_h] [threads]"""
import sys, tempfile, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config

variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
cfg = get_config(variant)


def clock(fn, reps=20):
    for _ in range(8):          # every lane has had its first pass (workspaces, code objects) before the clock starts
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    view = api.ImageView(synthetic_image(0), api.Channels.rgba)
    seg = api.Segmentation.process(view, env)
    print(f"{variant} lanes {api.ext.lane_count(env)}")
    print(f"process (1 image)            {clock(lambda: api.Segmentation.process(view, env)):7.2f} ms")
    print(f"compute_mask (1 point)       {clock(lambda: seg.compute_mask(api.Point(512, 512))):7.2f} ms")
    print(f"compute_masks (3 masks)      {clock(lambda: seg.compute_masks(api.Point(512, 512))):7.2f} ms")
    for n in (2, 4, 8, 16):
        views = [api.ImageView(synthetic_image(i), api.Channels.rgba) for i in range(n)]
        segs = api.Segmentation.process_batch(views, env)
        tp = clock(lambda: api.Segmentation.process_batch(views, env), 10)
        tm = clock(lambda: api.Segmentation.compute_mask_batch(segs, points=[api.Point(512, 512)] * n), 10)
        print(f"batch {n:2d}: process_batch {tp:7.2f} ms ({tp / n:5.2f}/img)  compute_mask_batch {tm:6.2f} ms ({tm / n:5.2f}/img)"
              f"  -> {1e3 * n / (tp + tm):6.1f} img/s")
    src = np.ascontiguousarray(synthetic_image(0))
    dst = np.empty_like(src)
    print(f"host memcpy 4 MiB            {clock(lambda: np.copyto(dst, src)):7.3f} ms")
