"""TEST INFRASTRUCTURE — CPU restatement of the pre/post-processing around dlimgedit's BiRefNet call
(`segment_objects`).  Only tests may import this; the product path never does.

The network itself is an ONNX graph run through onnxruntime in the reference (absent here, out of scope); what the
library computes itself is restated:
  prepare_image  /root/reference/src/segmentation.cpp:244-256   u8 HWC -> f32 NCHW, (x/255 - mean) / std
  process_mask   /root/reference/src/segmentation.cpp:258-270   u8 = uint8_t(sigmoid(x) * 255.f), sigmoid segmentation.hpp:87
  resize_mask    /root/reference/src/image.cpp:53-62            oracle/stb_resize.py::resize_mask
Pinned by the reference's known-answer tests for the first two (test/test_segmentation.cpp:152-180 ->
tests/test_oracle_kats.py); resize_mask has no upstream vector (parity unpinned, see stb_resize.py).
"""
from __future__ import annotations

import numpy as np

from .stb_resize import resize_mask  # noqa: F401  (re-exported: the third step of the post-processing)

f32 = np.float32

BIREFNET_MEAN = (0.485, 0.456, 0.406)      # segmentation.cpp:231-232
BIREFNET_STD = (0.229, 0.224, 0.225)


def prepare_image(pixels: np.ndarray, mean=BIREFNET_MEAN, std=BIREFNET_STD) -> np.ndarray:
    """u8 [H,W,C>=3] -> f32 [1,3,H,W]; float arithmetic in the reference's order: value = x / 255.f, (value - mean) / std."""
    pixels = np.asarray(pixels, dtype=np.uint8)
    out = np.empty((1, 3) + pixels.shape[:2], f32)
    for c in range(3):
        value = (pixels[:, :, c].astype(f32) / f32(255)).astype(f32)
        out[0, c] = ((value - f32(mean[c])).astype(f32) / f32(std[c])).astype(f32)
    return out


def sigmoid(x: np.ndarray) -> np.ndarray:
    """1.0f / (1.0f + std::exp(-x)) in float; the float exp is taken as the correctly rounded value (glibc's expf is
    within 0.502 ulp of it, i.e. equal except when the true value sits on a rounding boundary)."""
    x = np.asarray(x, dtype=f32)
    with np.errstate(over="ignore"):              # exp(104) -> inf in float, as in the reference
        e = np.exp(-x.astype(np.float64)).astype(f32)
    return (f32(1) / (f32(1) + e).astype(f32)).astype(f32)


def process_mask(logits: np.ndarray) -> np.ndarray:
    """f32 [H,W] (the graph's output 0, channel 0) -> u8 [H,W] = uint8_t(sigmoid(x) * 255.f) (truncation)."""
    v = (sigmoid(logits) * f32(255)).astype(f32)
    return v.astype(np.int32).astype(np.uint8)
