"""CPU ORACLE (test infrastructure) for dlimg::resize -- the stb_image_resize call behind
ResizeLongestSide::resize (/root/reference/src/image.cpp:37-51, /root/reference/src/segmentation.cpp:60-70):

    stbir_resize_uint8_generic(..., num_channels, STBIR_ALPHA_CHANNEL_NONE, flags 0, STBIR_EDGE_CLAMP,
                               STBIR_FILTER_DEFAULT, STBIR_COLORSPACE_SRGB, nullptr)

stb_image_resize is an un-vendored dependency (pinned at nothings/stb @ 5736b15f, depend/stb/CMakeLists.txt:3-7;
that commit carries stb_image_resize.h v0.97); its source is not in the image, so this restates the
published algorithm of that version:

  * default filter: Catmull-Rom when an axis is upsampled (scale > 1), Mitchell-Netravali (B = C = 1/3)
    otherwise; support 2 for both;
  * sRGB: every channel (alpha channel = NONE) goes u8 -> linear float through a 256-entry table and back
    through Giesen's float -> sRGB8 table conversion;
  * per axis, per-pixel contributor lists with float coefficients: upsampling gathers (coefficients
    normalised to sum 1), downsampling scatters kernel(x)*scale from every input pixel incl. the clamped
    margin and normalises per output pixel;
  * horizontal pass first (into float rows of the output width), then vertical; plain float
    multiply-add in increasing source order; edges clamp.

Pinned only by the reference's single KAT for this function (test/test_image.cpp:51-69, restated in
tests/test_oracle_kats.py); listed as an unpinned assumption in DESIGN.md otherwise.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------------------------- sRGB

def srgb_to_linear_table() -> np.ndarray:
    """stbir__srgb_uchar_to_linear_float: the literal table holds the exact curve printed with 6 decimals."""
    out = np.empty(256, dtype=np.float64)
    for i in range(256):
        c = i / 255.0
        v = c / 12.92 if c <= 0.04045 else ((c + 0.055) / 1.055) ** 2.4
        out[i] = float(f"{v:.6f}")
    return out.astype(f32)


# fp32_to_srgb8_tab4 of stbir__linear_to_srgb_uchar (F. Giesen's float->sRGB8 conversion)
_TAB4 = np.array([
    0x0073000d, 0x007a000d, 0x0080000d, 0x0087000d, 0x008d000d, 0x0094000d, 0x009a000d, 0x00a1000d,
    0x00a7001a, 0x00b4001a, 0x00c1001a, 0x00ce001a, 0x00da001a, 0x00e7001a, 0x00f4001a, 0x0101001a,
    0x010e0033, 0x01280033, 0x01410033, 0x015b0033, 0x01750033, 0x018f0033, 0x01a80033, 0x01c20033,
    0x01dc0067, 0x020f0067, 0x02430067, 0x02760067, 0x02aa0067, 0x02dd0067, 0x03110067, 0x03440067,
    0x037800ce, 0x03df00ce, 0x044600ce, 0x04ad00ce, 0x051400ce, 0x057b00c5, 0x05dd00bc, 0x063b00b5,
    0x06970158, 0x07420142, 0x07e30130, 0x087b0120, 0x090b0112, 0x09940106, 0x0a1700fc, 0x0a9500f2,
    0x0b0f01cb, 0x0bf401ae, 0x0ccb0195, 0x0d950180, 0x0e56016e, 0x0f0d015e, 0x0fbc0150, 0x10630143,
    0x11070264, 0x1238023e, 0x1357021d, 0x14660201, 0x156601e9, 0x165a01d3, 0x174401c0, 0x182401af,
    0x18fe0331, 0x1a9602fe, 0x1c1502d2, 0x1d7e02ad, 0x1ed4028d, 0x201a0270, 0x21520256, 0x227d0240,
    0x239f0443, 0x25c003fe, 0x27bf03c4, 0x29a10392, 0x2b6a0367, 0x2d1d0341, 0x2ebe031f, 0x304d0300,
    0x31d105b0, 0x34a80555, 0x37520507, 0x39d504c5, 0x3c37048b, 0x3e7c0458, 0x40a8042a, 0x42bd0401,
    0x44c20798, 0x488e071e, 0x4c1c06b6, 0x4f76065d, 0x52a50610, 0x55ac05cc, 0x5892058f, 0x5b590559,
    0x5e0c0a23, 0x631c0980, 0x67db08f6, 0x6c55087f, 0x70940818, 0x74a007bd, 0x787d076c, 0x7c330723,
], dtype=np.uint32)

_MINVAL_BITS = np.uint32((127 - 13) << 23)
_ALMOST_ONE_BITS = np.uint32(0x3f7fffff)


def linear_to_srgb_uchar(x: np.ndarray) -> np.ndarray:
    """stbir__linear_to_srgb_uchar on an fp32 array -> uint8."""
    x = np.asarray(x, dtype=f32)
    minval = _MINVAL_BITS.view(f32)
    almost = _ALMOST_ONE_BITS.view(f32)
    x = np.where(x > minval, x, minval)          # NaN -> minval, as the C tests are written
    x = np.where(x > almost, almost, x)
    u = x.view(np.uint32)
    tab = _TAB4[(u - _MINVAL_BITS) >> np.uint32(20)]
    bias = (tab >> np.uint32(16)) << np.uint32(9)
    scale = tab & np.uint32(0xffff)
    t = (u >> np.uint32(12)) & np.uint32(0xff)
    return ((bias + scale * t) >> np.uint32(16)).astype(np.uint8)


# ------------------------------------------------------------------------------------------ filters

def _catmullrom(x: f32) -> f32:
    x = f32(abs(x))
    if x < f32(1):
        return f32(1) - x * x * (f32(2.5) - f32(1.5) * x)
    if x < f32(2):
        return f32(2) - x * (f32(4) + x * (f32(0.5) * x - f32(2.5)))
    return f32(0)


def _mitchell(x: f32) -> f32:
    x = f32(abs(x))
    if x < f32(1):
        return (f32(16) + x * x * (f32(21) * x - f32(36))) / f32(18)
    if x < f32(2):
        return (f32(32) + x * (f32(-60) + x * (f32(36) - f32(7) * x))) / f32(18)
    return f32(0)


def _trapezoid(x: f32, s: f32) -> f32:
    """STBIR_FILTER_BOX: stbir__filter_trapezoid(x, s), s <= 1."""
    half = f32(s / f32(2))
    t = f32(f32(0.5) + half)
    x = f32(abs(x))
    if x >= t:
        return f32(0)
    r = f32(f32(0.5) - half)
    if x <= r:
        return f32(1)
    return f32(f32(t - x) / s)


def axis_contributors(in_size: int, out_size: int, box: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Gather-form contributor table of one axis: (first[out], count[out], coef[out, max_taps]).
    Source indices first..first+count-1 may fall outside [0, in_size): they clamp to the edge.
    box=False: STBIR_FILTER_DEFAULT (Catmull-Rom up, Mitchell down, support 2); box=True: STBIR_FILTER_BOX
    (trapezoid, support 0.5 + s/2 with s = 1/scale when upsampling and scale when downsampling)."""
    scale = f32(out_size) / f32(in_size)
    up = scale > 1                                  # stbir__use_upsampling
    filter_scale = f32(f32(1) / scale) if up else scale
    support = f32(f32(0.5) + f32(filter_scale / f32(2))) if box else f32(2)
    up_kernel = (lambda x: _trapezoid(x, filter_scale)) if box else _catmullrom
    down_kernel = (lambda x: _trapezoid(x, filter_scale)) if box else _mitchell
    lists: List[List[Tuple[int, f32]]] = [[] for _ in range(out_size)]
    if up:
        radius = support * scale                    # out_filter_radius
        for n in range(out_size):
            center = f32(n) + f32(0.5)
            lo = (center - radius) / scale
            hi = (center + radius) / scale
            in_center = center / scale
            first = int(math.floor(float(lo) + 0.5))
            last = int(math.floor(float(hi) - 0.5))
            coefs = []
            i = 0
            while i <= last - first:
                c = up_kernel(in_center - (f32(i + first) + f32(0.5)))
                if i == 0 and c == 0:               # leading zero: drop the pixel
                    first += 1
                    continue
                coefs.append(c)
                i += 1
            total = f32(0)
            for c in coefs:
                total = f32(total + c)
            fs = f32(1) / total
            coefs = [f32(c * fs) for c in coefs]
            while coefs and coefs[-1] == 0:         # trailing zeros carry no weight
                coefs.pop()
            lists[n] = [(first + k, c) for k, c in enumerate(coefs)]
    else:
        in_radius = support / scale                 # in_pixels_radius
        margin = int(math.ceil(float(support * f32(2) / scale))) // 2      # filter_pixel_margin
        # scatter form: every input pixel (incl. margin) -> a few output pixels
        scat = []
        for j in range(-margin, in_size + margin):
            center = f32(j) + f32(0.5)
            lo = (center - in_radius) * scale
            hi = (center + in_radius) * scale
            out_center = center * scale
            first = int(math.floor(float(lo) + 0.5))
            last = int(math.floor(float(hi) - 0.5))
            cs = [f32(down_kernel((f32(i) + f32(0.5)) - out_center) * scale) for i in range(first, last + 1)]
            scat.append((j, first, cs))
        totals = [f32(0)] * out_size
        for j, first, cs in scat:                   # per-output normalisation, summed in input order
            for k, c in enumerate(cs):
                i = first + k
                if 0 <= i < out_size:
                    totals[i] = f32(totals[i] + c)
        for j, first, cs in scat:
            for k, c in enumerate(cs):
                i = first + k
                if 0 <= i < out_size:
                    w = f32(c * (f32(1) / totals[i]))
                    if w != 0:
                        lists[i].append((j, w))
    taps = max(len(l) for l in lists)
    first = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    coef = np.zeros((out_size, taps), f32)
    for n, l in enumerate(lists):
        idx = [j for j, _ in l]
        assert idx == list(range(idx[0], idx[0] + len(idx))), "contributors must be contiguous"
        first[n], count[n] = idx[0], len(idx)
        coef[n, :len(idx)] = [c for _, c in l]
    return first, count, coef


# -------------------------------------------------------------------------------------------- resize

def _apply_axis(src: np.ndarray, first, count, coef, axis: int) -> np.ndarray:
    """Weighted gather along `axis` of an fp32 array, accumulating in source order (mul then add)."""
    n_in = src.shape[axis]
    out_shape = list(src.shape)
    out_shape[axis] = len(first)
    acc = np.zeros(out_shape, f32)
    for t in range(coef.shape[1]):
        idx = np.clip(first + t, 0, n_in - 1)
        w = np.where(t < count, coef[:, t], f32(0)).astype(f32)
        taken = np.take(src, idx, axis=axis)
        shape = [1] * src.ndim
        shape[axis] = len(first)
        acc = (acc + taken * w.reshape(shape)).astype(f32)
    return acc


def resize_srgb(pixels: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """u8 [H,W,C] (or [H,W]) -> u8 [out_h,out_w,C], the restated stbir_resize_uint8_generic call."""
    pixels = np.asarray(pixels, dtype=np.uint8)
    squeeze = pixels.ndim == 2
    if squeeze:
        pixels = pixels[:, :, None]
    h, w, _ = pixels.shape
    lin = srgb_to_linear_table()[pixels]                         # decode
    hf, hc, hk = axis_contributors(w, out_w)
    vf, vc, vk = axis_contributors(h, out_h)
    rows = _apply_axis(lin, hf, hc, hk, axis=1)                  # horizontal first
    full = _apply_axis(rows, vf, vc, vk, axis=0)
    out = linear_to_srgb_uchar(full)
    return out[:, :, 0] if squeeze else out


def resize_mask(mask: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """u8 [H,W] -> u8 [out_h,out_w]: stbir_resize_uint8_generic(1 channel, EDGE_CLAMP, FILTER_BOX, COLORSPACE_LINEAR),
    the call of dlimg::resize_mask (/root/reference/src/image.cpp:53-62).  Linear colour space: decode x / 255,
    encode (int)(saturate(v) * 255 + 0.5).  The reference holds no known-answer test for this call: parity unpinned
    beyond the contributor machinery shared with resize_srgb (which its KAT pins)."""
    mask = np.asarray(mask, dtype=np.uint8)
    h, w = mask.shape
    lin = (mask.astype(f32) / f32(255)).astype(f32)
    hf, hc, hk = axis_contributors(w, out_w, box=True)
    vf, vc, vk = axis_contributors(h, out_h, box=True)
    rows = _apply_axis(lin, hf, hc, hk, axis=1)
    full = _apply_axis(rows, vf, vc, vk, axis=0)
    sat = np.clip(full, f32(0), f32(1)).astype(f32)
    return (sat * f32(255) + f32(0.5)).astype(f32).astype(np.int32).astype(np.uint8)
