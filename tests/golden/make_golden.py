"""Generates the committed golden vectors of tests/golden/*.npz.

Runs ONLY in the build container: it needs Hugging Face `transformers` (SamModel), the one
independent implementation of the published SAM model that is importable there.
The vectors pin the CPU oracle (oracle/sam_oracle.py); the reference's own golden masks are
git-LFS stubs in the checkout and cannot be used.

    python tests/golden/make_golden.py [--full] [--vit-l] [--vit-h] [--only NAME]

What is stored (all small, strided samples where tensors are big):
  sam_<variant>.npz   image seed, prompt, embedding samples, low-res logit samples, all four IoU predictions,
                      packed final masks of mask tokens 1..3 -- all produced by HF SamModel + torch F.interpolate.
                      Which of the three the single-mask mode returns is NOT stored: the tests restate the
                      selection rule on HF's IoU predictions themselves.
  sam_vit_h_1800x1200.npz   BASELINE config 5's geometry: a 1800x1200 RGB image (resized to 1024x683 by the stb
                      restatement oracle/stb_resize.py, which the reference's resize KAT pins), five Halton
                      point prompts in original coordinates (rounded into the resized frame as the reference
                      does, src/segmentation.cpp:26,146), HF on the padded input, torch post-processing with crop
  post_torch.npz      bilinear post-processing of a fixed logit plane by torch.nn.functional.interpolate
"""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from conftest import halton_points, synthetic_image  # noqa: E402
from dlimgedit_amd import weights as W  # noqa: E402
from dlimgedit_amd.sam_config import get_config  # noqa: E402
from oracle import sam_oracle as O  # noqa: E402

from transformers import SamConfig, SamMaskDecoderConfig, SamModel, SamVisionConfig  # noqa: E402

OUT = Path(__file__).resolve().parent
EMB_STRIDE = 257        # sample every 257th value of the 4096x256 embedding
LOW_STRIDE = 61         # sample every 61st value of each 256x256 logit plane


def hf_model(cfg, params):
    vc = SamVisionConfig(hidden_size=cfg.embed_dim, num_hidden_layers=cfg.depth, num_attention_heads=cfg.num_heads,
                         global_attn_indexes=list(cfg.global_attn_indexes), mlp_dim=cfg.mlp_dim)
    # norm1..4 of the two-way blocks: Meta's decoder (what the reference's graphs are exported from) builds them with
    # nn.LayerNorm's default eps 1e-5; HF reads it from the config, whose default is 1e-6
    dc = SamMaskDecoderConfig(layer_norm_eps=O.DEC_LN_EPS)
    model = SamModel(SamConfig(vision_config=vc, mask_decoder_config=dc)).eval()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in W.to_hf_state_dict(cfg, params).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("mask_embed" in m for m in missing), (missing, unexpected)
    return model


def torch_post(low, h, w):
    t = F.interpolate(torch.from_numpy(low)[None, None], size=(1024, 1024), mode="bilinear", align_corners=False)
    s = np.float32(1024.0) / np.float32(max(h, w))
    ph, pw = int(np.float32(h) * s + np.float32(0.5)), int(np.float32(w) * s + np.float32(0.5))
    t = F.interpolate(t[..., :ph, :pw], size=(h, w), mode="bilinear", align_corners=False)
    return t[0, 0].numpy()


def prompt_outputs(model, emb, prompt):
    """All four low-res masks / IoU predictions of the decoder for one prompt (token 0 first)."""
    with torch.no_grad():
        o3 = model(image_embeddings=emb, multimask_output=True, **prompt)
        o1 = model(image_embeddings=emb, multimask_output=False, **prompt)
    low = torch.cat([o1.pred_masks[0, 0], o3.pred_masks[0, 0]], 0).numpy()
    iou = torch.cat([o1.iou_scores[0, 0], o3.iou_scores[0, 0]], 0).numpy()
    return low, iou


def make_variant(variant, seed, image_seed, compare_oracle=True):
    cfg = get_config(variant)
    params = W.synthetic_weights(cfg, seed)
    model = hf_model(cfg, params)
    img = synthetic_image(image_seed)
    x = O.preprocess(O.create_image_tensor(img, O.CH_RGBA))
    with torch.no_grad():
        emb = model.get_image_embeddings(torch.from_numpy(x)[None])
    pt = dict(input_points=torch.tensor([[[[512., 512.]]]]), input_labels=torch.tensor([[[1]]]))
    bx = dict(input_boxes=torch.tensor([[[256., 256., 768., 768.]]]))
    emb_tok = emb[0].reshape(256, -1).T.contiguous().numpy()
    out = {"seed": seed, "image_seed": image_seed, "emb_samples": emb_tok.reshape(-1)[::EMB_STRIDE].copy()}
    for name, prompt in (("point", pt), ("box", bx)):
        low, iou = prompt_outputs(model, emb, prompt)
        out[f"{name}_low_samples"] = low.reshape(4, -1)[:, ::LOW_STRIDE].copy()
        out[f"{name}_iou"] = iou
        out[f"{name}_masks_bits"] = np.stack([np.packbits(torch_post(low[t], 1024, 1024) > 0) for t in (1, 2, 3)])
    np.savez_compressed(OUT / f"sam_{variant}.npz", **out)
    if compare_oracle:      # report how the oracle compares right now
        oe = O.encode_image(x, params, cfg)
        print(variant, "oracle vs HF embedding max-abs", float(np.abs(oe - emb_tok).max()))


def make_nonsquare(variant, seed, w, h, n_prompts=5):
    """Longest side != 1024 (BASELINE config 5): resize -> pad -> encode -> prompts rounded into the resized frame ->
    crop + second bilinear back to w x h."""
    from oracle import stb_resize
    cfg = get_config(variant)
    params = W.synthetic_weights(cfg, seed)
    model = hf_model(cfg, params)
    img = synthetic_image(w, width=w, height=h, channels=4)[:, :, :3].copy()
    rs = O.ResizeLongestSide()
    rw, rh = rs.target_extent(w, h)
    resized = stb_resize.resize_srgb(img, rw, rh)
    x = O.preprocess(O.create_image_tensor(resized, O.CH_RGB))
    with torch.no_grad():
        emb = model.get_image_embeddings(torch.from_numpy(x)[None])
    emb_tok = emb[0].reshape(256, -1).T.contiguous().numpy()
    pts = halton_points(n_prompts, w, h)
    out = {"seed": seed, "width": w, "height": h, "points": np.array(pts, np.int32),
           "emb_samples": emb_tok.reshape(-1)[::EMB_STRIDE].copy()}
    lows, ious, masks = [], [], []
    for (px, py) in pts:
        tx, ty = rs.transform(px, py)
        prompt = dict(input_points=torch.tensor([[[[float(tx), float(ty)]]]]), input_labels=torch.tensor([[[1]]]))
        low, iou = prompt_outputs(model, emb, prompt)
        lows.append(low.reshape(4, -1)[:, ::LOW_STRIDE].copy())
        ious.append(iou)
        masks.append(np.stack([np.packbits(torch_post(low[t], h, w) > 0) for t in (1, 2, 3)]))
    out["low_samples"], out["iou"], out["masks_bits"] = np.stack(lows), np.stack(ious), np.stack(masks)
    np.savez_compressed(OUT / f"sam_{variant}_{w}x{h}.npz", **out)


def make_post():
    rng = np.random.default_rng(123)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    low = (3 * np.sin(xx * 0.07 + 0.5) * np.cos(yy * 0.05) + rng.normal(0, 0.3, (256, 256))).astype(np.float32)
    out = {"low": low.astype(np.float16)}     # stored as f16 (exactly representable input for both sides)
    low = low.astype(np.float16).astype(np.float32)
    for (h, w) in [(1024, 1024), (1200, 1800), (683, 1024), (512, 512), (37, 91)]:
        t = torch_post(low, h, w)
        out[f"bits_{h}x{w}"] = np.packbits(t > 0)
        out[f"samples_{h}x{w}"] = t.reshape(-1)[::97].copy()
    np.savez_compressed(OUT / "post_torch.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    want = lambda name: only is None or only == name   # noqa: E731
    if want("vit_test"):
        make_variant("vit_test", seed=7, image_seed=0)
    if want("vit_test80"):
        make_variant("vit_test80", seed=7, image_seed=4)
    if "--full" in sys.argv and want("vit_b"):
        make_variant("vit_b", seed=0, image_seed=0)
    if "--vit-l" in sys.argv and want("vit_l"):     # a minute or two of CPU time, ~4 GB of memory
        make_variant("vit_l", seed=0, image_seed=0, compare_oracle=False)
    if "--vit-h" in sys.argv:       # several minutes of CPU time and ~10 GB of memory each
        if want("vit_h"):
            make_variant("vit_h", seed=0, image_seed=0, compare_oracle=False)
        if want("vit_h_1800x1200"):
            make_nonsquare("vit_h", seed=0, w=1800, h=1200)
    if want("post"):
        make_post()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")
