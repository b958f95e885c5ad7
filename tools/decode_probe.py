"""compute_mask in a loop on one cached embedding: target for rocprofv3 --kernel-trace (decoder kernel breakdown).
python tools/decode_probe.py [variant] [reps] [prompts per call]"""
import sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config
variant = sys.argv[1] if len(sys.argv) > 1 else "vit_test"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = get_config(variant)
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    seg = api.Segmentation.process(api.ImageView(synthetic_image(0), api.Channels.rgba), env)
    import time
    for _ in range(3):
        seg.compute_mask(api.Point(512, 512))
    t0 = time.perf_counter()
    for i in range(reps):
        if P == 1:
            seg.compute_mask(api.Point(100 + i, 512))
        else:
            api.Segmentation.compute_mask_batch([seg] * P, points=[api.Point(100 + i + 7 * j, 512) for j in range(P)])
    print(f"{1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call, {P} prompt(s)")
