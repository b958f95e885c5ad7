"""Shared fixtures.  `-m "not gpu"` runs on the CPU-only build box; `-m gpu` needs one MI355X."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synthetic_image(seed: int, width: int = 1024, height: int = 1024, channels: int = 4) -> np.ndarray:
    """SURVEY.md §8(d) synthetic input: low-frequency sinusoids + noise, alpha 255."""
    rng = np.random.default_rng(1000 + seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.zeros((height, width, channels), np.uint8)
    for c in range(min(3, channels)):
        f = np.full((height, width), 128.0, np.float32)
        for _ in range(8):
            fx, fy = rng.uniform(0.002, 0.02, 2)
            ph = rng.uniform(0, 6.28)
            f += 14.0 * np.sin(xx * fx + yy * fy + ph).astype(np.float32)
        f += rng.uniform(-8, 8, (height, width)).astype(np.float32)
        img[:, :, c] = np.clip(f, 0, 255).astype(np.uint8)
    if channels == 4:
        img[:, :, 3] = 255
    return img


def halton_points(n: int, width: int, height: int, start: int = 1):
    """n point prompts on the Halton sequence (bases 2, 3) inside a width x height image."""
    def radical(i, base):
        f, r = 1.0, 0.0
        while i > 0:
            f /= base
            r += f * (i % base)
            i //= base
        return r
    return [(min(width - 1, int(radical(i, 2) * width)), min(height - 1, int(radical(i, 3) * height)))
            for i in range(start, start + n)]


@pytest.fixture(scope="session")
def model_dirs(tmp_path_factory):
    """variant -> (model directory with seeded synthetic weights, params dict, config); built lazily."""
    from dlimgedit_amd import weights as W
    from dlimgedit_amd.sam_config import get_config
    cache = {}

    def get(variant: str, seed: int = 7):
        key = (variant, seed)
        if key not in cache:
            cfg = get_config(variant)
            d = tmp_path_factory.mktemp(f"models_{variant}_{seed}")
            params = W.write_synthetic_model_dir(d, cfg, seed)
            cache[key] = (str(d), params, cfg)
        return cache[key]

    return get


# ---- parity tolerances (fp32 oracle / Hugging Face fixtures vs the f16-operand MFMA path with fp32 accumulation, softmax
# and LayerNorm statistics).  Set at 3-5x what the kernels measure on MI355X (round 3: embedding 0.0036-0.0048, low-res
# logits 0.006, IoU predictions 0.0004), so a kernel that loses a few bits of accuracy fails; every checked value is also
# appended to gpurun_out/parity_margins.txt so the margins can be read after a run.
EMB_TOL = 0.015        # max-abs error of the image embedding (LayerNorm'ed values, |x| ~ 4)
LOGIT_TOL = 0.03       # max-abs error of the low-res mask logits (std ~ 1.3, range about +-5)
IOU_PRED_TOL = 0.003   # max-abs error of the decoder's IoU predictions
IOU_BAR = 0.98         # BASELINE.json: mask IoU vs the CPU reference


def within(name: str, err, tol: float) -> float:
    """assert err < tol, and leave (name, err, tol) in gpurun_out/parity_margins.txt."""
    err = float(err)
    try:
        out = ROOT / "gpurun_out"
        out.mkdir(exist_ok=True)
        with open(out / "parity_margins.txt", "a") as f:
            f.write(f"{name}\t{err:.6g}\t{tol:.6g}\n")
    except OSError:
        pass
    assert err < tol, f"{name}: {err:.6g} is not below {tol:.6g}"
    return err


def at_least(name: str, value, bar: float) -> float:
    """assert value >= bar (IoU-like quantities), logged like within()."""
    value = float(value)
    try:
        out = ROOT / "gpurun_out"
        out.mkdir(exist_ok=True)
        with open(out / "parity_margins.txt", "a") as f:
            f.write(f"{name}\t{value:.6g}\t>={bar:.6g}\n")
    except OSError:
        pass
    assert value >= bar, f"{name}: {value:.6g} is below {bar:.6g}"
    return value


def iou(a: np.ndarray, b: np.ndarray) -> float:
    a, b = a > 0, b > 0
    union = np.logical_or(a, b).sum()
    return 1.0 if union == 0 else float(np.logical_and(a, b).sum()) / float(union)


def single_mask_index(iou4) -> int:
    """The single-mask decoder graph's choice, restated independently of the oracle: SamOnnxModel.select_masks adds
    (num_points - 2.5) * [1000, 0, 0, 0] to the four IoU predictions and takes the argmax; dlimgedit always sends two
    prompt tokens (src/segmentation.cpp:146-152), so token 0 is pushed down by 500 and the winner is the best of
    tokens 1..3 (the first one on ties, as argmax does)."""
    score = np.asarray(iou4, np.float32) + np.float32(2 - 2.5) * np.array([1000, 0, 0, 0], np.float32)
    return int(np.argmax(score))
