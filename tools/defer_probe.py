"""process() that waits for its encoder pass itself (DLIMGEDIT_SYNC_PROCESS=1, the reference's timing) against the default,
which leaves the wait to the first query of the handle (csrc/segmentation.hpp): one synchronous caller of slots 3 + 4, three
interleaved rounds of 3 s, then slot 3 alone in a loop from one thread.  ViT-B, synthetic weights, host buffers in and out.
    gpurun -- 'python3 tools/defer_probe.py > gpurun_out/defer_probe.txt'"""
import sys, time, tempfile, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config
cfg = get_config("vit_b")
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    view = api.ImageView(synthetic_image(0), api.Channels.rgba)
    ref = api.Segmentation.process(view, env).compute_mask(api.Point(512, 512))
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["DLIMGEDIT_SYNC_PROCESS"] = mode
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                m = api.Segmentation.process(view, env).compute_mask(api.Point(512, 512)); n += 1
            dt = time.perf_counter() - t0
            assert (m == ref).all()
            print(f"round {rnd} DLIMGEDIT_SYNC_PROCESS={mode}: {n / dt:7.1f} images/s one synchronous caller", flush=True)
    # process() alone from one thread
    for mode in ("0", "1"):
        os.environ["DLIMGEDIT_SYNC_PROCESS"] = mode
        segs, t0 = [], time.perf_counter()
        for i in range(400):
            segs.append(api.Segmentation.process(view, env))
            if len(segs) > 16: segs.pop(0).close()
        for s in segs: api.ext.get_embedding(s); s.close()
        print(f"process() only, one thread, DLIMGEDIT_SYNC_PROCESS={mode}: {400 / (time.perf_counter() - t0):7.1f} images/s", flush=True)
