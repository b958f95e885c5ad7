"""One-off full-size parity run on the GPU box: python tools/parity_full.py vit_b|vit_l|vit_h
Encodes one synthetic image through the drop-in API and the CPU oracle; prints embedding / logit errors,
mask IoU (point, box, 3-mask mode) and foreground fractions."""
import sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image, iou
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config
from oracle import sam_oracle as O

variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
cfg = get_config(variant)
with tempfile.TemporaryDirectory() as d:
    t0 = time.time(); params = W.write_synthetic_model_dir(d, cfg, seed=0); print(f"weights {time.time()-t0:.1f}s", flush=True)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    img = synthetic_image(0)
    t0 = time.time(); seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env); print(f"gpu first process {time.time()-t0:.2f}s", flush=True)
    t0 = time.time(); seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env); print(f"gpu second process {1e3*(time.time()-t0):.2f} ms (host API path incl. 4 MiB H2D + sync)", flush=True)
    t0 = time.time(); ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA); print(f"oracle encode {time.time()-t0:.1f}s", flush=True)
    emb = api.ext.get_embedding(seg)
    print("embedding max-abs err", float(np.abs(emb - ora.embedding).max()), "rms", float(np.sqrt(((emb-ora.embedding)**2).mean())))
    for name, kw_gpu, kw_ora in (("point", dict(point=api.Point(512, 512)), dict(point=(512, 512))),
                                 ("point2", dict(point=api.Point(200, 800)), dict(point=(200, 800))),
                                 ("box", dict(region=api.Region(api.Point(256, 256), api.Point(768, 768))), dict(region=(256, 256, 768, 768)))):
        lg, ig = api.ext.get_logits(seg, **kw_gpu)
        lo, io = ora.logits(**kw_ora)
        prompt = list(kw_gpu.values())[0]
        t0 = time.time(); mg = seg.compute_mask(prompt); dt = time.time() - t0
        mo = ora.compute_mask(**kw_ora)
        print(f"{name}: logits max-abs err {np.abs(lg-lo).max():.4f} (std {lo.std():.3f}) iou-pred err {np.abs(ig-io).max():.4f} "
              f"mask IoU {iou(mg, mo):.5f} fg {float((mo>0).mean()):.3f} differing px {int((mg!=mo).sum())}  compute_mask {1e3*dt:.2f} ms", flush=True)
    masks = seg.compute_masks(api.Point(512, 512)); om, oa = ora.compute_masks((512, 512))
    print("3-mask IoUs", [round(iou(m.image, o), 5) for m, o in zip(masks, om)], "fg", [round(float((o>0).mean()), 3) for o in om])
