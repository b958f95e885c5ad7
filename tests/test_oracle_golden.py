"""Oracle vs the committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py
from Hugging Face SamModel and torch.nn.functional.interpolate in the build container)."""
from pathlib import Path

import numpy as np
import pytest

from conftest import single_mask_index, synthetic_image
from dlimgedit_amd import weights as W
from dlimgedit_amd.sam_config import get_config
from oracle import sam_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
EMB_STRIDE, LOW_STRIDE = 257, 61


def _check_variant(variant):
    g = np.load(GOLD / f"sam_{variant}.npz")
    cfg = get_config(variant)
    params = W.synthetic_weights(cfg, int(g["seed"]))
    img = synthetic_image(int(g["image_seed"]))
    seg = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
    emb = seg.embedding.reshape(-1)[::EMB_STRIDE]
    assert np.abs(emb - g["emb_samples"]).max() < 2e-4
    for name, kw in (("point", dict(point=(512, 512))), ("box", dict(region=(256, 256, 768, 768)))):
        low, iou = seg.logits(**kw)
        assert np.abs(low.reshape(4, -1)[:, ::LOW_STRIDE] - g[f"{name}_low_samples"]).max() < 5e-4
        assert np.abs(iou - g[f"{name}_iou"]).max() < 1e-4
        # the rule applied to HF's own predictions picks the same token as the oracle's select_single on its own
        best = single_mask_index(g[f"{name}_iou"])
        assert best in (1, 2, 3) and O.select_single(iou, 2) == best
        masks = np.unpackbits(g[f"{name}_masks_bits"], axis=1).reshape(3, 1024, 1024) * 255
        # logits agree to ~1e-5, so only pixels whose logit is within that of zero may differ
        assert (seg.compute_mask(**kw) != masks[best - 1]).mean() < 2e-5
        if name == "point":     # multi-mask mode: decoder outputs 1..3 with their IoU predictions
            multi, acc = seg.compute_masks(kw["point"])
            for t in range(3):
                assert (multi[t] != masks[t]).mean() < 2e-5
                assert abs(acc[t] - float(g[f"{name}_iou"][t + 1])) < 1e-4


def test_oracle_matches_hf_reduced_variant():
    _check_variant("vit_test")


def test_oracle_matches_hf_head_dim_80_variant():
    _check_variant("vit_test80")


def test_oracle_matches_hf_vit_b():
    """Full ViT-B (the benchmark model): ~25 s of numpy on 8 cores."""
    _check_variant("vit_b")


@pytest.mark.parametrize("h,w", [(1024, 1024), (1200, 1800), (683, 1024), (512, 512), (37, 91)])
def test_postprocess_matches_torch_interpolate(h, w):
    g = np.load(GOLD / "post_torch.npz")
    low = g["low"].astype(np.float32)
    mine = O.postprocess_logits(low, (h, w))
    assert np.abs(mine.reshape(-1)[::97] - g[f"samples_{h}x{w}"]).max() < 5e-5
    want = np.unpackbits(g[f"bits_{h}x{w}"])[:h * w].reshape(h, w).astype(bool)
    assert ((mine > 0) != want).mean() < 2e-5
