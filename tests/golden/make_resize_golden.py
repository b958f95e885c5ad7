"""Generates tests/golden/resize_upsample_pil.npz: an UP-sampling vector for the longest-side resize that does not
come from oracle/stb_resize.py.

The reference's only resize KAT (test/test_image.cpp:51-69) is a 2x Mitchell down-sample; the Catmull-Rom up-sampling half
of stb_image_resize's default filter has no reference-held vector.  This pins the filter FAMILY and the sRGB handling
against an independent implementation: Pillow's BICUBIC (Keys cubic with a = -0.5 = Catmull-Rom) applied in linear light,
with the sRGB transfer function evaluated from its defining formula in float64.  It is not bit-exact to stb (Pillow
truncates and renormalises the kernel at the borders where stb clamps, and stb goes through its float tables), so the
tests compare within a stated tolerance away from the borders.

    python tests/golden/make_resize_golden.py        (build container only: needs Pillow)
"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image  # noqa: E402

OUT = Path(__file__).resolve().parent / "resize_upsample_pil.npz"


def srgb_decode(u8):
    c = u8.astype(np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def srgb_encode(lin):
    lin = np.clip(lin.astype(np.float64), 0.0, 1.0)
    c = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * lin ** (1.0 / 2.4) - 0.055)
    return np.clip(np.floor(c * 255.0 + 0.5), 0, 255).astype(np.uint8)


def pil_bicubic_linear_light(img, out_w, out_h):
    planes = []
    for c in range(img.shape[2]):
        lin = srgb_decode(img[:, :, c]).astype(np.float32)
        up = Image.fromarray(lin, mode="F").resize((out_w, out_h), Image.BICUBIC)
        planes.append(np.asarray(up, dtype=np.float32))
    lin_out = np.stack(planes, axis=2)
    return srgb_encode(lin_out), lin_out


def main():
    cases = {}
    # (name, source w, h, channels, out w, out h): longest side scaled as ResizeLongestSide would (to a small "1024")
    for name, w, h, ch, ow, oh in (("rgb_96x60_to_256x160", 96, 60, 3, 256, 160),
                                   ("rgba_50x80_to_160x256", 50, 80, 4, 160, 256)):
        src = synthetic_image(77 + w, width=w, height=h, channels=4)
        if ch == 4:   # a varying fourth channel: STBIR_ALPHA_CHANNEL_NONE resamples it like a colour channel
            yy, xx = np.mgrid[0:h, 0:w]
            src[:, :, 3] = np.clip(40 + 3 * xx + 2 * yy, 0, 255).astype(np.uint8)
        src = np.ascontiguousarray(src[:, :, :ch])
        out, _ = pil_bicubic_linear_light(src, ow, oh)
        cases[f"{name}_src"] = src
        cases[f"{name}_out"] = out
    np.savez_compressed(OUT, **cases)
    print("wrote", OUT, {k: v.shape for k, v in cases.items()})


if __name__ == "__main__":
    main()
