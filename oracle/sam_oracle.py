"""CPU ORACLE for the Segment-Anything hot path of dlimgedit  --  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy fp32, what `Segmentation::process()` and
`Segmentation::compute_mask()` compute in the reference.  It is the checker the HIP path is
compared against; it is never imported by the product (`dlimgedit_amd/`), only by `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.

What is restated from where
---------------------------
* host glue (C++ in the reference):
    resize rule / coordinate rounding   /root/reference/src/segmentation.cpp:26,60-74
    create_image_tensor                 /root/reference/src/segmentation.cpp:81-106
    prompt packing                      /root/reference/src/segmentation.cpp:131-152
    write_mask_image (threshold)        /root/reference/src/segmentation.cpp:108-116
    single/multi mask selection         /root/reference/src/segmentation.cpp:154-173
* model arithmetic (ONNX graphs, NOT present under /root/reference):
    the graphs are exports of Meta's Segment Anything (`segment_anything` `Sam` /
    `SamOnnxModel`, exporter arguments at /root/reference/script/export_models.py:21-43).
    Their published algorithm is restated here: in-graph preprocessing (normalise, zero-pad to
    1024, HWC->CHW), ViT image encoder with windowed/global attention and decomposed relative
    position bias, neck, prompt encoder, two-way-transformer mask decoder, `SamOnnxModel`
    mask selection (`iou + (num_points-2.5)*[1000,0,0,0]`) and `mask_postprocessing`
    (bilinear 256->1024, crop to the pre-padding size, bilinear to the original size).

Pinning (DESIGN.md "Oracle")
----------------------------
* the reference's own known-answer tests for the glue (test/test_segmentation.cpp:15-99) are
  restated in tests/test_oracle_kats.py and pass against this file;
* the model arithmetic is pinned against Hugging Face `transformers` SamModel (third-party
  re-implementation of the same published model, importable in the build container only) on
  seeded weights: tests/golden/make_golden.py generated the committed vectors;
* the reference's golden masks are git-LFS stubs in this checkout and pin MobileSAM weights that
  are not available: end-to-end parity with the reference's *binary* is therefore unpinned and
  DESIGN.md says so.

Numerics: everything is IEEE fp32, evaluated in the order written (numpy does not contract
mul+add into fma), so the pixel pre/post stages are bit-reproducible by the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

try:  # scipy is present in the image; fall back to math.erf for portability
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float32])

f32 = np.float32

IMAGE_SIZE = 1024
PIXEL_MEAN = np.array([123.675, 116.28, 103.53], dtype=f32)
PIXEL_STD = np.array([58.395, 57.12, 57.375], dtype=f32)

# dlimg::Channels  (/root/reference/src/include/dlimgedit/dlimgedit.hpp:29)
CH_MASK, CH_RGB, CH_RGBA, CH_BGRA, CH_ARGB = 1, 3, 4, 5, 6


def channel_count(channels: int) -> int:
    """dlimgedit.impl.hpp:15 -- bgra/argb are 4-byte formats."""
    return 4 if channels > 4 else channels


# =============================================================================================
# host glue

def scale_coord(coord: int, scale) -> int:
    """segmentation.cpp:26  int(coord * scale + 0.5f), all in fp32."""
    return int(f32(f32(coord) * f32(scale)) + f32(0.5))


class ResizeLongestSide:
    """segmentation.cpp:58-74."""

    def __init__(self, max_side: int = IMAGE_SIZE):
        self.max_side = max_side
        self.original = (0, 0)     # (width, height)
        self.scale = f32(1)

    def target_extent(self, width: int, height: int) -> Tuple[int, int]:
        self.original = (width, height)
        self.scale = f32(self.max_side) / f32(max(width, height))
        if self.scale != 1:
            return scale_coord(width, self.scale), scale_coord(height, self.scale)
        return width, height

    def transform(self, x: int, y: int) -> Tuple[int, int]:
        return scale_coord(x, self.scale), scale_coord(y, self.scale)


def create_image_tensor(pixels: np.ndarray, channels: int) -> np.ndarray:
    """segmentation.cpp:81-106: u8 [H,W,C] -> f32 [H,W,3] with the channel map; alpha dropped,
    no scaling.  `pixels` is the tightly packed image (the reference ignores the stride here)."""
    cmap = {CH_MASK: (0, 0, 0), CH_BGRA: (2, 1, 0), CH_ARGB: (1, 2, 3)}.get(channels, (0, 1, 2))
    pixels = np.asarray(pixels, dtype=np.uint8)
    if pixels.ndim == 2:
        pixels = pixels[:, :, None]
    return pixels[:, :, list(cmap)].astype(f32)


def write_mask_image(logits: np.ndarray, index: int, extent: Tuple[int, int]) -> np.ndarray:
    """segmentation.cpp:108-116: strict `> 0` -> 255, cropped to extent (w,h) with the tensor's
    own row stride.  logits: [1,K,H,W]."""
    w, h = extent
    return np.where(logits[0, index, :h, :w] > 0, 255, 0).astype(np.uint8)


def pack_prompt(rs: ResizeLongestSide, point=None, region=None):
    """segmentation.cpp:135-152 -> (coords f32 [2,2], labels f32 [2])."""
    assert (point is None) != (region is None)
    if point is not None:
        px, py = rs.transform(*point)
        coords = np.array([[px, py], [0, 0]], dtype=f32)   # Point(0,0) transforms to (0,0)
        labels = np.array([1, -1], dtype=f32)
    else:
        x0, y0 = rs.transform(region[0], region[1])
        x1, y1 = rs.transform(region[2], region[3])
        coords = np.array([[x0, y0], [x1, y1]], dtype=f32)
        labels = np.array([2, 3], dtype=f32)
    return coords, labels


# =============================================================================================
# in-graph preprocessing of the encoder (export_models.py:26 `use_preprocess=True`)

def preprocess(image_hw3: np.ndarray, image_size: int = IMAGE_SIZE) -> np.ndarray:
    """f32 [H',W',3] in 0..255 -> f32 [3,S,S]: (x-mean)/std, zero pad bottom/right, CHW."""
    h, w, _ = image_hw3.shape
    x = (image_hw3.astype(f32) - PIXEL_MEAN) / PIXEL_STD
    out = np.zeros((3, image_size, image_size), dtype=f32)
    out[:, :h, :w] = x.transpose(2, 0, 1)
    return out


def patchify(chw: np.ndarray, patch: int = 16) -> np.ndarray:
    """[3,S,S] -> [(S/p)^2, 3*p*p] rows ordered (py,px), columns (c,iy,ix) = conv-weight order."""
    c, s, _ = chw.shape
    g = s // patch
    x = chw.reshape(c, g, patch, g, patch).transpose(1, 3, 0, 2, 4)
    return np.ascontiguousarray(x.reshape(g * g, c * patch * patch))


# =============================================================================================
# building blocks

def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True, dtype=f32)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=f32)
    return (xc / np.sqrt(var + f32(eps))) * w + b


def gelu(x: np.ndarray) -> np.ndarray:
    return (f32(0.5) * x * (f32(1) + _erf(x * f32(0.7071067811865476)).astype(f32))).astype(f32)


def softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True, dtype=f32)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    y = x @ w.T
    return y if b is None else y + b


# =============================================================================================
# image encoder (SAM ViT)

def _rel_table(rel: np.ndarray, q: int, k: int) -> np.ndarray:
    """get_rel_pos for q==k and table length 2*q-1 (no interpolation): R[qi,ki] = rel[qi-ki+k-1]."""
    assert rel.shape[0] == 2 * max(q, k) - 1 and q == k
    idx = np.arange(q)[:, None] - np.arange(k)[None, :] + (k - 1)
    return rel[idx]                                           # [q,k,hd]


def attention_from_qkv(qkv: np.ndarray, rel_h: np.ndarray, rel_w: np.ndarray, heads: int, s: int) -> np.ndarray:
    """qkv: [nb, s*s, 3*D] (q | k | v, head-major) -> attention output [nb, s*s, D] before proj.
    S = (q*scale) k^T + q.Rh[qy-ky+s-1] + q.Rw[qx-kx+s-1] with the raw q in the bias terms."""
    nb, n, d3 = qkv.shape
    d = d3 // 3
    hd = d // heads
    qkv = qkv.reshape(nb, n, 3, heads, hd).transpose(2, 0, 3, 1, 4)        # [3,nb,h,n,hd]
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = f32(hd ** -0.5)
    rh = _rel_table(rel_h, s, s)
    rw = _rel_table(rel_w, s, s)
    out = np.empty((nb, heads, n, hd), dtype=f32)
    for b in range(nb):
        for h in range(heads):
            qq = q[b, h]
            attn = (qq * scale) @ k[b, h].T                                 # [n,n]
            r_q = qq.reshape(s, s, hd)
            rel_hh = np.einsum("hwc,hkc->hwk", r_q, rh, optimize=True)      # [s,s,kh]
            rel_ww = np.einsum("hwc,wkc->hwk", r_q, rw, optimize=True)      # [s,s,kw]
            attn = attn.reshape(s, s, s, s) + rel_hh[:, :, :, None] + rel_ww[:, :, None, :]
            out[b, h] = softmax(attn.reshape(n, n)) @ v[b, h]
    return out.transpose(0, 2, 1, 3).reshape(nb, n, d)


def _attention(x: np.ndarray, p: Dict[str, np.ndarray], pre: str, heads: int) -> np.ndarray:
    """x: [nb, S, S, D] (already LayerNorm'ed, windows or whole grid) -> same shape before proj."""
    nb, s, _, d = x.shape
    qkv = linear(x.reshape(nb, s * s, d), p[pre + ".qkv.w"], p[pre + ".qkv.b"])
    return attention_from_qkv(qkv, p[pre + ".rel_h"], p[pre + ".rel_w"], heads, s).reshape(nb, s, s, d)


def windowed_attention_from_qkv(qkv: np.ndarray, qkv_bias: np.ndarray, rel_h, rel_w, heads: int,
                                grid: int = 64, ws: int = 14) -> np.ndarray:
    """Token-order qkv [grid*grid, 3D] -> windowed attention output [grid*grid, D]; the zero-padded
    window tokens carry qkv == bias (their LayerNorm'ed input is zero)."""
    d3 = qkv.shape[-1]
    pad = (ws - grid % ws) % ws
    gp = grid + pad
    full = np.broadcast_to(qkv_bias.astype(f32), (gp, gp, d3)).copy()
    full[:grid, :grid] = qkv.reshape(grid, grid, d3)
    nw = gp // ws
    win = full.reshape(nw, ws, nw, ws, d3).transpose(0, 2, 1, 3, 4).reshape(nw * nw, ws * ws, d3)
    o = attention_from_qkv(win, rel_h, rel_w, heads, ws)
    d = d3 // 3
    o = o.reshape(nw, nw, ws, ws, d).transpose(0, 2, 1, 3, 4).reshape(gp, gp, d)
    return np.ascontiguousarray(o[:grid, :grid].reshape(grid * grid, d))


def _window_partition(x: np.ndarray, ws: int):
    g, _, d = x.shape
    pad = (ws - g % ws) % ws
    xp = np.pad(x, ((0, pad), (0, pad), (0, 0)))
    gp = g + pad
    nw = gp // ws
    win = xp.reshape(nw, ws, nw, ws, d).transpose(0, 2, 1, 3, 4).reshape(nw * nw, ws, ws, d)
    return win, gp


def _window_unpartition(win: np.ndarray, ws: int, gp: int, g: int) -> np.ndarray:
    nw = gp // ws
    d = win.shape[-1]
    x = win.reshape(nw, nw, ws, ws, d).transpose(0, 2, 1, 3, 4).reshape(gp, gp, d)
    return x[:g, :g]


def encoder_block(x: np.ndarray, p, i: int, cfg, return_attn: bool = False):
    """x: [g,g,D] fp32 residual stream."""
    pre = f"enc.L{i}"
    g = x.shape[0]
    h = layer_norm(x, p[pre + ".ln1.w"], p[pre + ".ln1.b"], 1e-6)
    if i in cfg.global_attn_indexes:
        a = _attention(h[None], p, pre, cfg.num_heads)[0]
    else:
        win, gp = _window_partition(h, cfg.window_size)       # zero pad AFTER LayerNorm
        a = _window_unpartition(_attention(win, p, pre, cfg.num_heads), cfg.window_size, gp, g)
    attn_out = a
    x = x + linear(a, p[pre + ".proj.w"], p[pre + ".proj.b"])
    h = layer_norm(x, p[pre + ".ln2.w"], p[pre + ".ln2.b"], 1e-6)
    h = gelu(linear(h, p[pre + ".fc1.w"], p[pre + ".fc1.b"]))
    x = x + linear(h, p[pre + ".fc2.w"], p[pre + ".fc2.b"])
    return (x, attn_out) if return_attn else x


def neck(x: np.ndarray, p, cfg) -> np.ndarray:
    """[g,g,D] -> [g,g,256]  (1x1 conv, LN2d, 3x3 conv pad 1, LN2d; no biases; eps 1e-6)."""
    g = x.shape[0]
    y = x @ p["enc.neck.conv1.w"].T
    y = layer_norm(y, p["enc.neck.ln1.w"], p["enc.neck.ln1.b"], 1e-6)
    yp = np.pad(y, ((1, 1), (1, 1), (0, 0)))
    w = p["enc.neck.conv2.w"]                                 # [co,ci,3,3]
    z = np.zeros((g, g, w.shape[0]), dtype=f32)
    for ky in range(3):
        for kx in range(3):
            z += yp[ky:ky + g, kx:kx + g] @ w[:, :, ky, kx].T
    return layer_norm(z, p["enc.neck.ln2.w"], p["enc.neck.ln2.b"], 1e-6)


def encode_image(chw: np.ndarray, p, cfg, taps: Optional[dict] = None) -> np.ndarray:
    """Preprocessed [3,S,S] -> embedding in token-major layout [g*g, 256]
    (the reference's NCHW 1x256x64x64 is `emb.T.reshape(256,64,64)`)."""
    g = cfg.grid
    x = patchify(chw, cfg.patch_size) @ p["enc.patch.w"].T + p["enc.patch.b"] + p["enc.pos"]
    x = x.reshape(g, g, cfg.embed_dim).astype(f32)
    if taps is not None:
        taps["patch"] = x.reshape(g * g, -1).copy()
    for i in range(cfg.depth):
        x = encoder_block(x, p, i, cfg)
        if taps is not None:
            taps[f"L{i}"] = x.reshape(g * g, -1).copy()
    e = neck(x, p, cfg).reshape(g * g, cfg.out_chans)
    return np.ascontiguousarray(e, dtype=f32)


# =============================================================================================
# prompt encoder + mask decoder

def _pe_encoding(coords01: np.ndarray, gauss: np.ndarray) -> np.ndarray:
    """coords in [0,1], (x,y) order -> [..., 256] = [sin, cos](2*pi*((2c-1) @ G))."""
    c = (f32(2) * coords01.astype(f32) - f32(1)) @ gauss
    c = f32(2 * np.pi) * c
    return np.concatenate([np.sin(c), np.cos(c)], axis=-1).astype(f32)


def image_pe(p, grid: int = 64) -> np.ndarray:
    """Dense positional encoding of the embedding grid, token-major [grid*grid, 256]."""
    t = (np.arange(grid, dtype=f32) + f32(0.5)) / f32(grid)
    yy, xx = np.meshgrid(t, t, indexing="ij")
    return _pe_encoding(np.stack([xx, yy], axis=-1), p["pe.gauss"]).reshape(grid * grid, -1)


def embed_prompt(coords: np.ndarray, labels: np.ndarray, p) -> np.ndarray:
    """SamOnnxModel._embed_points: coords [n,2] in resized-image pixels, labels [n] -> [n,256]."""
    c = (coords.astype(f32) + f32(0.5)) / f32(IMAGE_SIZE)
    e = _pe_encoding(c, p["pe.gauss"])
    lab = labels.astype(f32)[:, None]
    e = e * (lab != -1)
    e = e + p["pe.not_a_point"][None, :] * (lab == -1)
    for i in range(4):
        e = e + p["pe.point"][i][None, :] * (lab == i)
    return e.astype(f32)


def _dec_attention(q_in, k_in, v_in, p, pre: str) -> np.ndarray:
    q = linear(q_in, p[pre + ".q.w"], p[pre + ".q.b"])
    k = linear(k_in, p[pre + ".k.w"], p[pre + ".k.b"])
    v = linear(v_in, p[pre + ".v.w"], p[pre + ".v.b"])
    heads = 8
    hd = q.shape[-1] // heads
    qh = q.reshape(-1, heads, hd).transpose(1, 0, 2)
    kh = k.reshape(-1, heads, hd).transpose(1, 0, 2)
    vh = v.reshape(-1, heads, hd).transpose(1, 0, 2)
    a = softmax((qh @ kh.transpose(0, 2, 1)) * f32(hd ** -0.5))
    o = (a @ vh).transpose(1, 0, 2).reshape(q.shape[0], heads * hd)
    return linear(o, p[pre + ".o.w"], p[pre + ".o.b"])


# LayerNorm eps of the mask decoder.  Meta's TwoWayAttentionBlock / TwoWayTransformer build norm1..4 and
# norm_final_attn with nn.LayerNorm's default (1e-5) -- the graphs the reference runs are exports of that code
# (/root/reference/script/export_models.py:29-43); only LayerNorm2d and the encoder blocks use 1e-6.  (Hugging Face's
# port takes the block norms' eps from SamMaskDecoderConfig.layer_norm_eps, default 1e-6: tests/golden/make_golden.py
# sets it to 1e-5 so that both sides state the same model.)
DEC_LN_EPS = 1e-5


def two_way_transformer(tokens: np.ndarray, src: np.ndarray, pos: np.ndarray, p):
    """tokens [T,256] (also the query PE), src/pos [4096,256] -> (queries, keys)."""
    queries, keys, qpe = tokens, src, tokens
    for i in range(2):
        pre = f"dec.L{i}"
        if i == 0:
            queries = _dec_attention(queries, queries, queries, p, pre + ".self")
        else:
            qq = queries + qpe
            queries = queries + _dec_attention(qq, qq, queries, p, pre + ".self")
        queries = layer_norm(queries, p[pre + ".ln1.w"], p[pre + ".ln1.b"], DEC_LN_EPS)
        queries = queries + _dec_attention(queries + qpe, keys + pos, keys, p, pre + ".t2i")
        queries = layer_norm(queries, p[pre + ".ln2.w"], p[pre + ".ln2.b"], DEC_LN_EPS)
        h = np.maximum(linear(queries, p[pre + ".mlp.fc1.w"], p[pre + ".mlp.fc1.b"]), 0)
        queries = queries + linear(h, p[pre + ".mlp.fc2.w"], p[pre + ".mlp.fc2.b"])
        queries = layer_norm(queries, p[pre + ".ln3.w"], p[pre + ".ln3.b"], DEC_LN_EPS)
        keys = keys + _dec_attention(keys + pos, queries + qpe, queries, p, pre + ".i2t")
        keys = layer_norm(keys, p[pre + ".ln4.w"], p[pre + ".ln4.b"], DEC_LN_EPS)
    queries = queries + _dec_attention(queries + qpe, keys + pos, keys, p, "dec.final")
    queries = layer_norm(queries, p["dec.ln_final.w"], p["dec.ln_final.b"], DEC_LN_EPS)
    return queries.astype(f32), keys.astype(f32)


def _mlp3(x, p, pre):
    x = np.maximum(linear(x, p[pre + ".0.w"], p[pre + ".0.b"]), 0)
    x = np.maximum(linear(x, p[pre + ".1.w"], p[pre + ".1.b"]), 0)
    return linear(x, p[pre + ".2.w"], p[pre + ".2.b"])


def upscale(keys: np.ndarray, p, grid: int = 64) -> np.ndarray:
    """[4096,256] -> [256*256, 32] pixel-major: ConvT2x2/2 -> LN2d(eps 1e-6) -> GELU -> ConvT2x2/2 -> GELU."""
    w1, b1 = p["dec.up1.w"], p["dec.up1.b"]                  # [256,64,2,2]
    c1 = w1.shape[1]
    y = keys @ w1.reshape(w1.shape[0], -1)                    # [4096, 64*4] columns (co,dy,dx)
    y = y.reshape(grid, grid, c1, 2, 2).transpose(0, 3, 1, 4, 2).reshape(2 * grid, 2 * grid, c1) + b1
    y = gelu(layer_norm(y, p["dec.up_ln.w"], p["dec.up_ln.b"], 1e-6))
    w2, b2 = p["dec.up2.w"], p["dec.up2.b"]                  # [64,32,2,2]
    c2 = w2.shape[1]
    z = y.reshape(-1, c1) @ w2.reshape(c1, -1)
    z = z.reshape(2 * grid, 2 * grid, c2, 2, 2).transpose(0, 3, 1, 4, 2).reshape(4 * grid, 4 * grid, c2) + b2
    return gelu(z).reshape(-1, c2).astype(f32)


def decode_masks(emb: np.ndarray, coords: np.ndarray, labels: np.ndarray, p, taps: Optional[dict] = None):
    """Embedding [4096,256] + packed prompt -> (low-res logits [4,256,256], iou [4])."""
    sparse = embed_prompt(coords, labels, p)
    tokens = np.concatenate([p["dec.iou_token"][None, :], p["dec.mask_tokens"], sparse], axis=0).astype(f32)
    src = emb + p["pe.no_mask"][None, :]                      # has_mask_input == 0
    pos = image_pe(p)
    queries, keys = two_way_transformer(tokens, src, pos, p)
    up = upscale(keys, p)                                     # [65536,32]
    hyper = np.stack([_mlp3(queries[1 + m], p, f"dec.hyper{m}") for m in range(4)], axis=0)   # [4,32]
    masks = (hyper @ up.T).reshape(4, 256, 256).astype(f32)
    iou = _mlp3(queries[0], p, "dec.iou").astype(f32)
    if taps is not None:
        taps.update(tokens=tokens, queries=queries, keys=keys, up=up, hyper=hyper)
    return masks, iou


def select_single(iou: np.ndarray, num_points: int) -> int:
    """SamOnnxModel.select_masks: argmax(iou + (num_points-2.5)*[1000,0,0,0])."""
    score = iou.astype(f32).copy()
    score[0] = score[0] + f32(num_points - 2.5) * f32(1000)
    return int(np.argmax(score))


# =============================================================================================
# mask post-processing (in-graph in the reference; fixed fp32 evaluation order, no fma)

def _lin_coeffs(out_size: int, in_size: int):
    """torch/ONNX half-pixel bilinear, align_corners=False: indices i0,i1 and weight of i1."""
    scale = f32(in_size) / f32(out_size)
    d = np.arange(out_size, dtype=f32)
    src = scale * (d + f32(0.5)) - f32(0.5)
    src = np.maximum(src, f32(0))
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    l1 = (src - i0.astype(f32)).astype(f32)
    l1 = np.minimum(np.maximum(l1, f32(0)), f32(1))
    return i0, i1, (f32(1) - l1).astype(f32), l1


def bilinear_resize(x: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """x [H,W] f32 -> [out_h,out_w]:  out = ly0*(lx0*a00 + lx1*a01) + ly1*(lx0*a10 + lx1*a11)."""
    h, w = x.shape
    y0, y1, wy0, wy1 = _lin_coeffs(out_h, h)
    x0, x1, wx0, wx1 = _lin_coeffs(out_w, w)
    top = x[y0][:, x0] * wx0[None, :] + x[y0][:, x1] * wx1[None, :]
    bot = x[y1][:, x0] * wx0[None, :] + x[y1][:, x1] * wx1[None, :]
    return (wy0[:, None] * top + wy1[:, None] * bot).astype(f32)


def postprocess_logits(lowres: np.ndarray, orig_hw: Tuple[int, int], image_size: int = IMAGE_SIZE) -> np.ndarray:
    """SamOnnxModel.mask_postprocessing: [256,256] -> [H,W] logits at original resolution."""
    h, w = orig_hw
    up = bilinear_resize(lowres, image_size, image_size)
    scale = f32(image_size) / f32(max(h, w))
    ph = int(math.floor(float(f32(scale * f32(h)) + f32(0.5))))
    pw = int(math.floor(float(f32(scale * f32(w)) + f32(0.5))))
    up = up[:ph, :pw]
    return bilinear_resize(up, h, w)


# =============================================================================================
# end-to-end, mirroring SegmentationImpl

class OracleSegmentation:
    """SegmentationImpl (segmentation.cpp:118-174) on the oracle."""

    def __init__(self, params, cfg):
        self.p, self.cfg = params, cfg
        self.rs = ResizeLongestSide(IMAGE_SIZE)
        self.embedding = None

    def process(self, pixels: np.ndarray, channels: int = CH_RGBA, resized: Optional[np.ndarray] = None):
        """pixels u8 [H,W,C].  If the longest side is not 1024 the caller supplies `resized`
        (the stb resize of config 5 is restated separately in oracle/stb_resize.py)."""
        h, w = pixels.shape[:2]
        tw, th = self.rs.target_extent(w, h)
        if (tw, th) != (w, h):
            if resized is None:
                from oracle.stb_resize import resize_srgb
                resized = resize_srgb(pixels, tw, th)
            assert resized.shape[:2] == (th, tw)
            pixels = resized
        x = preprocess(create_image_tensor(pixels, channels))
        self.embedding = encode_image(x, self.p, self.cfg)
        return self

    def logits(self, point=None, region=None):
        coords, labels = pack_prompt(self.rs, point, region)
        return decode_masks(self.embedding, coords, labels, self.p)

    def compute_mask(self, point=None, region=None) -> np.ndarray:
        """Single-mask mode (result_masks[1] == nullptr)."""
        low, iou = self.logits(point, region)
        w, h = self.rs.original
        best = select_single(iou, 2)
        full = postprocess_logits(low[best], (h, w))
        return write_mask_image(full[None, None], 0, (w, h))

    def compute_masks(self, point):
        """Multi-mask mode: decoder outputs 1..3 and their predicted IoU."""
        low, iou = self.logits(point=point)
        w, h = self.rs.original
        masks = [write_mask_image(postprocess_logits(low[i + 1], (h, w))[None, None], 0, (w, h)) for i in range(3)]
        return masks, [float(iou[i + 1]) for i in range(3)]
