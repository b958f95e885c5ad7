"""The decoder's image-side projection riding in the token self-attention's launch (DLIMGEDIT_DECODER_RIDE=1, default) against the
two launches on their own (=0): child processes, three interleaved rounds; bits of masks, logits and IoU predictions hashed.
    gpurun -- 'python3 tools/ride_probe.py > gpurun_out/ride_probe.txt'"""
import os, subprocess, sys, hashlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from conftest import synthetic_image
    from dlimgedit_amd import api, weights as W
    from dlimgedit_amd.sam_config import get_config
    d = "/tmp/ride_model"
    if not os.path.exists(d + "/segmentation"):
        os.makedirs(d, exist_ok=True); W.write_synthetic_model_dir(d, get_config("vit_b"), seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    view = api.ImageView(synthetic_image(0), api.Channels.rgba)
    seg = api.Segmentation.process(view, env)
    pts5 = [api.Point(100 + 150 * k, 200 + 100 * k) for k in range(5)]
    pts16 = [api.Point(40 + 60 * k, 900 - 50 * k) for k in range(16)]
    hs = []
    for _ in range(10):
        m = seg.compute_mask(api.Point(512, 512))
        five = api.Segmentation.compute_mask_batch([seg] * 5, points=pts5)
        many = api.Segmentation.compute_mask_batch([seg] * 16, points=pts16)
        three = seg.compute_masks(api.Point(300, 400))
        low, iou = api.ext.get_logits(seg, point=api.Point(512, 512))
        hs.append(hashlib.sha1(m.tobytes() + b"".join(f.tobytes() for f in five + many) + b"".join(t.image.tobytes() for t in three) + low.tobytes() + iou.tobytes()).hexdigest()[:12])
    def rate(fn, secs=2.0):
        fn(); n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < secs: fn(); n += 1
        return n / (time.perf_counter() - t0)
    r1 = rate(lambda: seg.compute_mask(api.Point(512, 512)))
    r5 = 5 * rate(lambda: api.Segmentation.compute_mask_batch([seg] * 5, points=pts5))
    r16 = 16 * rate(lambda: api.Segmentation.compute_mask_batch([seg] * 16, points=pts16))
    rs = rate(lambda: api.Segmentation.process(view, env).compute_mask(api.Point(512, 512)), 3.0)
    print(f"DLIMGEDIT_DECODER_RIDE={os.environ.get('DLIMGEDIT_DECODER_RIDE', '1')}: one prompt {r1:7.1f}/s ({1e3 / r1:.4f} ms), five per call {r5:7.1f} prompts/s, "
          f"sixteen per call {r16:7.1f} prompts/s, one synchronous caller {rs:6.1f} images/s, bits {sorted(set(hs))}", flush=True)
else:
    for rnd in range(3):
        for v in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DLIMGEDIT_DECODER_RIDE=v), check=True)
