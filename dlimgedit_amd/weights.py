"""Weight store for the SAM path: parameter inventory, seeded synthetic weights, the on-disk
format the HIP executor loads, and conversion from a Hugging Face / Meta style state dict.

The reference locates its models at `<model_directory>/segmentation/<name>.onnx`
(/root/reference/src/session.cpp:79-83).  This build keeps the directory convention and replaces
the ONNX graphs by one flat tensor file `<model_directory>/segmentation/sam_<variant>.dlw`:

    bytes 0..7    magic  b"DLIMGSAM"
    u32           version (1)
    u32           n_tensors
    i32[16]       config: embed_dim, depth, num_heads, mlp_dim, n_global, global_idx[8], pad[3]
    n_tensors x { char name[64]; u32 dtype(0=f32); u32 ndim; u64 dims[4]; u64 offset; u64 nbytes }
    tensor data, each tensor 64-byte aligned, little-endian f32

No real checkpoint is reachable from this environment (no network), so benchmarks and
parity tests use weights drawn from a counter-based generator: every value is a pure function of
(seed, tensor name, element index), identical on every machine and independent of numpy's stream.
"""
from __future__ import annotations

import struct
import zlib
from pathlib import Path
from typing import Dict, Iterable, List, Tuple

import numpy as np

from .sam_config import (SamConfig, DEC_DIM, DEC_MLP, DEC_DEPTH, DEC_DOWNSAMPLE,
                         NUM_MASK_TOKENS, IOU_HIDDEN)

MAGIC = b"DLIMGSAM"
VERSION = 1

# ---------------------------------------------------------------------------------------------
# parameter inventory

# init kinds: ("lin", fan_in) uniform with var 1/fan_in; ("u", a) uniform(-a,a); ("ln_w",) 1+u(0.1)
Spec = Tuple[str, Tuple[int, ...], Tuple]


def _attn_specs(prefix: str, dim: int, inner: int) -> List[Spec]:
    out = []
    for p, (o, i) in (("q", (inner, dim)), ("k", (inner, dim)), ("v", (inner, dim)), ("o", (dim, inner))):
        out.append((f"{prefix}.{p}.w", (o, i), ("lin", i)))
        out.append((f"{prefix}.{p}.b", (o,), ("u", 0.05)))
    return out


def _ln_specs(prefix: str, dim: int) -> List[Spec]:
    return [(f"{prefix}.w", (dim,), ("ln_w",)), (f"{prefix}.b", (dim,), ("u", 0.1))]


def param_specs(cfg: SamConfig) -> List[Spec]:
    d, hd, g, w = cfg.embed_dim, cfg.head_dim, cfg.grid, cfg.window_size
    k_patch = 3 * cfg.patch_size ** 2
    s: List[Spec] = [
        ("enc.patch.w", (d, k_patch), ("lin", k_patch)),
        ("enc.patch.b", (d,), ("u", 0.05)),
        ("enc.pos", (g * g, d), ("u", 0.5)),
    ]
    for i in range(cfg.depth):
        span = g if i in cfg.global_attn_indexes else w
        p = f"enc.L{i}"
        s += _ln_specs(f"{p}.ln1", d)
        s += [(f"{p}.qkv.w", (3 * d, d), ("lin", d)), (f"{p}.qkv.b", (3 * d,), ("u", 0.05)),
              (f"{p}.rel_h", (2 * span - 1, hd), ("u", 0.2)), (f"{p}.rel_w", (2 * span - 1, hd), ("u", 0.2)),
              (f"{p}.proj.w", (d, d), ("lin", d)), (f"{p}.proj.b", (d,), ("u", 0.05))]
        s += _ln_specs(f"{p}.ln2", d)
        s += [(f"{p}.fc1.w", (cfg.mlp_dim, d), ("lin", d)), (f"{p}.fc1.b", (cfg.mlp_dim,), ("u", 0.05)),
              (f"{p}.fc2.w", (d, cfg.mlp_dim), ("lin", cfg.mlp_dim)), (f"{p}.fc2.b", (d,), ("u", 0.05))]
    oc = cfg.out_chans
    s += [("enc.neck.conv1.w", (oc, d), ("lin", d))]
    s += _ln_specs("enc.neck.ln1", oc)
    s += [("enc.neck.conv2.w", (oc, oc, 3, 3), ("lin", oc * 9))]
    s += _ln_specs("enc.neck.ln2", oc)

    # prompt encoder (mask-input branch is never taken: has_mask_input == 0,
    # /root/reference/src/segmentation.cpp:43-45)
    s += [("pe.gauss", (2, DEC_DIM // 2), ("u", 1.7)),
          ("pe.point", (4, DEC_DIM), ("u", 0.5)),
          ("pe.not_a_point", (DEC_DIM,), ("u", 0.5)),
          ("pe.no_mask", (DEC_DIM,), ("u", 0.5))]

    # mask decoder
    s += [("dec.iou_token", (DEC_DIM,), ("u", 0.5)),
          ("dec.mask_tokens", (NUM_MASK_TOKENS, DEC_DIM), ("u", 0.5))]
    inner = DEC_DIM // DEC_DOWNSAMPLE
    for i in range(DEC_DEPTH):
        p = f"dec.L{i}"
        s += _attn_specs(f"{p}.self", DEC_DIM, DEC_DIM)
        s += _ln_specs(f"{p}.ln1", DEC_DIM)
        s += _attn_specs(f"{p}.t2i", DEC_DIM, inner)
        s += _ln_specs(f"{p}.ln2", DEC_DIM)
        s += [(f"{p}.mlp.fc1.w", (DEC_MLP, DEC_DIM), ("lin", DEC_DIM)), (f"{p}.mlp.fc1.b", (DEC_MLP,), ("u", 0.05)),
              (f"{p}.mlp.fc2.w", (DEC_DIM, DEC_MLP), ("lin", DEC_MLP)), (f"{p}.mlp.fc2.b", (DEC_DIM,), ("u", 0.05))]
        s += _ln_specs(f"{p}.ln3", DEC_DIM)
        s += _ln_specs(f"{p}.ln4", DEC_DIM)
        s += _attn_specs(f"{p}.i2t", DEC_DIM, inner)
    s += _attn_specs("dec.final", DEC_DIM, inner)
    s += _ln_specs("dec.ln_final", DEC_DIM)
    c1, c2 = DEC_DIM // 4, DEC_DIM // 8
    s += [("dec.up1.w", (DEC_DIM, c1, 2, 2), ("lin", DEC_DIM)), ("dec.up1.b", (c1,), ("u", 0.05))]
    s += _ln_specs("dec.up_ln", c1)
    s += [("dec.up2.w", (c1, c2, 2, 2), ("lin", c1)), ("dec.up2.b", (c2,), ("u", 0.05))]
    for m in range(NUM_MASK_TOKENS):
        dims = [(DEC_DIM, DEC_DIM), (DEC_DIM, DEC_DIM), (c2, DEC_DIM)]
        for j, (o, i) in enumerate(dims):
            s += [(f"dec.hyper{m}.{j}.w", (o, i), ("lin", i)), (f"dec.hyper{m}.{j}.b", (o,), ("u", 0.05))]
    dims = [(IOU_HIDDEN, DEC_DIM), (IOU_HIDDEN, IOU_HIDDEN), (NUM_MASK_TOKENS, IOU_HIDDEN)]
    for j, (o, i) in enumerate(dims):
        s += [(f"dec.iou.{j}.w", (o, i), ("lin", i)), (f"dec.iou.{j}.b", (o,), ("u", 0.05))]
    return s


# ---------------------------------------------------------------------------------------------
# counter-based generator (splitmix64 finaliser over (seed, crc32(name), index))

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix(x: np.ndarray) -> np.ndarray:
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def counter_uniform(seed: int, name: str, n: int) -> np.ndarray:
    """n floats in [-1, 1), a pure function of (seed, name, index); 24 random bits each."""
    with np.errstate(over="ignore"):
        key = np.uint64(seed) * _GOLD + np.uint64(zlib.crc32(name.encode()))
        key = _mix(np.array([key], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        bits = _mix((idx + np.uint64(1)) * _GOLD + key)
    u24 = (bits >> np.uint64(40)).astype(np.float64)          # 24 bits
    return (u24 * (2.0 / 16777216.0) - 1.0).astype(np.float32)


def synthetic_weights(cfg: SamConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    out: Dict[str, np.ndarray] = {}
    for name, shape, init in param_specs(cfg):
        n = int(np.prod(shape))
        u = counter_uniform(seed, name, n)
        if init[0] == "lin":
            a = np.float32(np.sqrt(3.0 / init[1]))
            v = u * a
        elif init[0] == "u":
            v = u * np.float32(init[1])
        elif init[0] == "ln_w":
            v = np.float32(1.0) + u * np.float32(0.1)
        else:  # pragma: no cover
            raise AssertionError(init)
        out[name] = np.ascontiguousarray(v.reshape(shape), dtype=np.float32)
    return out


def trained_like_weights(cfg: SamConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights with the activation statistics trained ViTs show and the uniform initialisation above
    does not: (1) a handful of "massive activation" channels of the residual stream, 50-300x the magnitude of the others
    and growing with depth (here: offsets in the patch-embedding bias plus contributions of every block's fc2 bias);
    (2) LayerNorm scales spread over two orders of magnitude (0.05 .. 5, log-uniform), with small scales on the
    massive channels as trained models learn them; (3) LayerNorm shifts of order one.  The folded-LayerNorm path
    (f16 copy of the raw stream, `acc - mean * colsum` in the consuming GEMM) is what this stresses."""
    p = synthetic_weights(cfg, seed)
    rng = np.random.default_rng(seed + 12345)
    D = cfg.embed_dim
    massive = rng.choice(D, size=4, replace=False)
    sign = rng.choice([-1.0, 1.0], size=4)
    # typical stream magnitude with these weights is ~1; the massive channels start at 50-120 and grow to 150-300
    p["enc.patch.b"] = p["enc.patch.b"].copy()
    p["enc.patch.b"][massive] += (sign * rng.uniform(50.0, 120.0, 4)).astype(np.float32)
    for i in range(cfg.depth):
        pre = f"enc.L{i}"
        b = p[pre + ".fc2.b"].copy()
        b[massive] += (sign * rng.uniform(5.0, 15.0, 4)).astype(np.float32)
        p[pre + ".fc2.b"] = b
        for ln in (".ln1", ".ln2"):
            g = np.exp(rng.uniform(np.log(0.05), np.log(5.0), D)).astype(np.float32)
            g[massive] = rng.uniform(0.02, 0.1, 4).astype(np.float32)
            p[pre + ln + ".w"] = g
            p[pre + ln + ".b"] = rng.uniform(-1.0, 1.0, D).astype(np.float32)
    return p


# ---------------------------------------------------------------------------------------------
# file format

def _cfg_block(cfg: SamConfig) -> bytes:
    gi = list(cfg.global_attn_indexes)
    if len(gi) > 8:
        raise ValueError("at most 8 global-attention layers are supported by the file header")
    vals = [cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_dim, len(gi)] + gi + [0] * (8 - len(gi)) + [0, 0, 0]
    return struct.pack("<16i", *vals)


def weight_file_name(cfg: SamConfig) -> str:
    return f"sam_{cfg.name}.dlw"


F16_MAX = 65504.0


def f16_operand(name: str) -> bool:
    """Tensors the device converts to f16 MFMA operands as they are (LayerNorm scales are folded into some of them at load
    time, which the C++ loader checks again after folding): the encoder's GEMM weights and rel-pos tables of the global
    blocks' kind, the neck convolutions, the decoder's image-side projections and up-scaling convolutions."""
    if name.endswith(".b"):
        return False
    if name.startswith("enc."):
        return name.endswith(".w") and ".ln" not in name or name.endswith((".rel_h", ".rel_w"))
    return name.endswith(".w") and any(t in name for t in (".k.w", ".v.w", ".q.w", "upscale", "conv"))


def check_f16_range(params: Dict[str, np.ndarray], allow_out_of_range: bool = False) -> None:
    """Real checkpoints are not guaranteed to fit this build's arithmetic: f16 MFMA operands overflow to infinity beyond
    65504, and an infinity turns every mask into a NaN pattern without any error.  Refuses non-finite values anywhere and
    out-of-range values in tensors that become f16 operands (`allow_out_of_range`: the writer's escape hatch for tests
    that plant such values to exercise the C++ loader, which refuses them again when the model is loaded)."""
    for name, arr in params.items():
        a = np.asarray(arr)
        if not np.isfinite(a).all():
            raise ValueError(f"{name}: holds non-finite values")
        if not allow_out_of_range and f16_operand(name) and a.size and float(np.abs(a).max()) > F16_MAX:
            raise ValueError(f"{name}: |value| up to {float(np.abs(a).max()):.6g} is outside the f16 range (65504) of the "
                             "MFMA operands this tensor becomes")


def save_weights(path, cfg: SamConfig, params: Dict[str, np.ndarray], allow_out_of_range: bool = False) -> Path:
    """Write `params` (must cover param_specs(cfg) exactly) to `path` in DLW v1."""
    path = Path(path)
    check_f16_range({s[0]: params[s[0]] for s in param_specs(cfg) if s[0] in params}, allow_out_of_range)
    specs = param_specs(cfg)
    names = [s[0] for s in specs]
    missing = [n for n in names if n not in params]
    if missing:
        raise ValueError(f"missing tensors: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    entry_size = 64 + 4 + 4 + 32 + 8 + 8
    header_size = 8 + 4 + 4 + 64 + entry_size * len(specs)
    offset = (header_size + 63) // 64 * 64
    table = []
    for name, shape, _ in specs:
        arr = params[name]
        if tuple(arr.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {arr.shape} != expected {shape}")
        nbytes = int(np.prod(shape)) * 4
        table.append((name, shape, offset, nbytes))
        offset = (offset + nbytes + 63) // 64 * 64
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(specs)))
        f.write(_cfg_block(cfg))
        for name, shape, off, nbytes in table:
            nb = name.encode()
            if len(nb) > 63:
                raise ValueError(f"tensor name too long: {name}")
            dims = list(shape) + [1] * (4 - len(shape))
            f.write(nb.ljust(64, b"\0"))
            f.write(struct.pack("<II4QQQ", 0, len(shape), *dims, off, nbytes))
        for name, shape, off, nbytes in table:
            f.seek(off)
            f.write(np.ascontiguousarray(params[name], dtype="<f4").tobytes())
        f.truncate(offset)
    return path


def load_weights(path) -> Tuple[Dict[str, int], Dict[str, np.ndarray]]:
    """Read a DLW file back (used by tests to check the writer against the C++ loader's view)."""
    raw = Path(path).read_bytes()
    if raw[:8] != MAGIC:
        raise ValueError("not a DLW weight file")
    version, n = struct.unpack_from("<II", raw, 8)
    if version != VERSION:
        raise ValueError(f"unsupported DLW version {version}")
    c = struct.unpack_from("<16i", raw, 16)
    meta = {"embed_dim": c[0], "depth": c[1], "num_heads": c[2], "mlp_dim": c[3],
            "global_attn_indexes": tuple(c[5:5 + c[4]])}
    out = {}
    pos = 80
    for _ in range(n):
        name = raw[pos:pos + 64].split(b"\0", 1)[0].decode()
        dtype, ndim, d0, d1, d2, d3, off, nbytes = struct.unpack_from("<II4QQQ", raw, pos + 64)
        pos += 64 + 4 + 4 + 32 + 16
        shape = (d0, d1, d2, d3)[:ndim]
        out[name] = np.frombuffer(raw, dtype="<f4", count=nbytes // 4, offset=off).reshape(shape).copy()
    return meta, out


def write_synthetic_model_dir(model_dir, cfg: SamConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Create `<model_dir>/segmentation/sam_<variant>.dlw` with seeded weights; returns them."""
    params = synthetic_weights(cfg, seed)
    save_weights(Path(model_dir) / "segmentation" / weight_file_name(cfg), cfg, params)
    return params


# ---------------------------------------------------------------------------------------------
# Hugging Face `SamModel` state-dict mapping (also the layout of converted Meta checkpoints)

def hf_name_map(cfg: SamConfig) -> List[Tuple[str, str]]:
    """(our name, HF SamModel state-dict key) for tensors that map one-to-one."""
    m = [("enc.patch.b", "vision_encoder.patch_embed.projection.bias"),
         ("enc.neck.ln1.w", "vision_encoder.neck.layer_norm1.weight"),
         ("enc.neck.ln1.b", "vision_encoder.neck.layer_norm1.bias"),
         ("enc.neck.conv2.w", "vision_encoder.neck.conv2.weight"),
         ("enc.neck.ln2.w", "vision_encoder.neck.layer_norm2.weight"),
         ("enc.neck.ln2.b", "vision_encoder.neck.layer_norm2.bias"),
         ("pe.gauss", "shared_image_embedding.positional_embedding"),
         ("pe.not_a_point", "prompt_encoder.not_a_point_embed.weight"),
         ("pe.no_mask", "prompt_encoder.no_mask_embed.weight"),
         ("dec.iou_token", "mask_decoder.iou_token.weight"),
         ("dec.mask_tokens", "mask_decoder.mask_tokens.weight"),
         ("dec.ln_final.w", "mask_decoder.transformer.layer_norm_final_attn.weight"),
         ("dec.ln_final.b", "mask_decoder.transformer.layer_norm_final_attn.bias"),
         ("dec.up1.w", "mask_decoder.upscale_conv1.weight"), ("dec.up1.b", "mask_decoder.upscale_conv1.bias"),
         ("dec.up2.w", "mask_decoder.upscale_conv2.weight"), ("dec.up2.b", "mask_decoder.upscale_conv2.bias"),
         ("dec.up_ln.w", "mask_decoder.upscale_layer_norm.weight"),
         ("dec.up_ln.b", "mask_decoder.upscale_layer_norm.bias")]
    for i in range(cfg.depth):
        a, b = f"enc.L{i}", f"vision_encoder.layers.{i}"
        m += [(f"{a}.ln1.w", f"{b}.layer_norm1.weight"), (f"{a}.ln1.b", f"{b}.layer_norm1.bias"),
              (f"{a}.qkv.w", f"{b}.attn.qkv.weight"), (f"{a}.qkv.b", f"{b}.attn.qkv.bias"),
              (f"{a}.rel_h", f"{b}.attn.rel_pos_h"), (f"{a}.rel_w", f"{b}.attn.rel_pos_w"),
              (f"{a}.proj.w", f"{b}.attn.proj.weight"), (f"{a}.proj.b", f"{b}.attn.proj.bias"),
              (f"{a}.ln2.w", f"{b}.layer_norm2.weight"), (f"{a}.ln2.b", f"{b}.layer_norm2.bias"),
              (f"{a}.fc1.w", f"{b}.mlp.lin1.weight"), (f"{a}.fc1.b", f"{b}.mlp.lin1.bias"),
              (f"{a}.fc2.w", f"{b}.mlp.lin2.weight"), (f"{a}.fc2.b", f"{b}.mlp.lin2.bias")]
    def attn(ours, theirs):
        r = []
        for p, q in (("q", "q_proj"), ("k", "k_proj"), ("v", "v_proj"), ("o", "out_proj")):
            r += [(f"{ours}.{p}.w", f"{theirs}.{q}.weight"), (f"{ours}.{p}.b", f"{theirs}.{q}.bias")]
        return r
    for i in range(DEC_DEPTH):
        a, b = f"dec.L{i}", f"mask_decoder.transformer.layers.{i}"
        m += attn(f"{a}.self", f"{b}.self_attn")
        m += attn(f"{a}.t2i", f"{b}.cross_attn_token_to_image")
        m += attn(f"{a}.i2t", f"{b}.cross_attn_image_to_token")
        for k in (1, 2, 3, 4):
            m += [(f"{a}.ln{k}.w", f"{b}.layer_norm{k}.weight"), (f"{a}.ln{k}.b", f"{b}.layer_norm{k}.bias")]
        m += [(f"{a}.mlp.fc1.w", f"{b}.mlp.lin1.weight"), (f"{a}.mlp.fc1.b", f"{b}.mlp.lin1.bias"),
              (f"{a}.mlp.fc2.w", f"{b}.mlp.lin2.weight"), (f"{a}.mlp.fc2.b", f"{b}.mlp.lin2.bias")]
    m += attn("dec.final", "mask_decoder.transformer.final_attn_token_to_image")
    hf_mlp = ("proj_in", "layers.0", "proj_out")
    for t in range(NUM_MASK_TOKENS):
        for j, h in enumerate(hf_mlp):
            m += [(f"dec.hyper{t}.{j}.w", f"mask_decoder.output_hypernetworks_mlps.{t}.{h}.weight"),
                  (f"dec.hyper{t}.{j}.b", f"mask_decoder.output_hypernetworks_mlps.{t}.{h}.bias")]
    for j, h in enumerate(hf_mlp):
        m += [(f"dec.iou.{j}.w", f"mask_decoder.iou_prediction_head.{h}.weight"),
              (f"dec.iou.{j}.b", f"mask_decoder.iou_prediction_head.{h}.bias")]
    return m


def to_hf_state_dict(cfg: SamConfig, params: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Our tensors laid out under HF `SamModel` keys (for cross-checking the oracle)."""
    d, g = cfg.embed_dim, cfg.grid
    sd = {hf: params[ours] for ours, hf in hf_name_map(cfg)}
    sd["vision_encoder.patch_embed.projection.weight"] = params["enc.patch.w"].reshape(d, 3, cfg.patch_size, cfg.patch_size)
    sd["vision_encoder.pos_embed"] = params["enc.pos"].reshape(1, g, g, d)
    sd["vision_encoder.neck.conv1.weight"] = params["enc.neck.conv1.w"].reshape(cfg.out_chans, d, 1, 1)
    for i in range(4):
        sd[f"prompt_encoder.point_embed.{i}.weight"] = params["pe.point"][i:i + 1]
    sd["prompt_encoder.not_a_point_embed.weight"] = params["pe.not_a_point"].reshape(1, -1)
    sd["prompt_encoder.no_mask_embed.weight"] = params["pe.no_mask"].reshape(1, -1)
    sd["mask_decoder.iou_token.weight"] = params["dec.iou_token"].reshape(1, -1)
    sd["prompt_encoder.shared_embedding.positional_embedding"] = params["pe.gauss"]
    return sd


def from_hf_state_dict(cfg: SamConfig, sd) -> Dict[str, np.ndarray]:
    """Inverse of `to_hf_state_dict` for real checkpoints (values may be torch tensors)."""
    def a(x):
        return np.ascontiguousarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32)
    d = cfg.embed_dim
    missing = [hf for _, hf in hf_name_map(cfg) if hf not in sd]
    if missing:
        raise ValueError(f"state dict lacks {len(missing)} tensors of {cfg.name} (wrong variant?), e.g. {missing[0]}")
    p = {ours: a(sd[hf]) for ours, hf in hf_name_map(cfg)}
    p["enc.patch.w"] = a(sd["vision_encoder.patch_embed.projection.weight"]).reshape(d, -1)
    p["enc.pos"] = a(sd["vision_encoder.pos_embed"]).reshape(-1, d)
    p["enc.neck.conv1.w"] = a(sd["vision_encoder.neck.conv1.weight"]).reshape(cfg.out_chans, d)
    p["pe.point"] = np.concatenate([a(sd[f"prompt_encoder.point_embed.{i}.weight"]) for i in range(4)], 0)
    for k in ("pe.not_a_point", "pe.no_mask", "dec.iou_token"):
        p[k] = p[k].reshape(-1)
    want = {n: s for n, s, _ in param_specs(cfg)}
    for n, s in want.items():
        if tuple(p[n].shape) != tuple(s):
            raise ValueError(f"{n}: checkpoint shape {p[n].shape} != {s}")
    return p


# ---------------------------------------------------------------------------------------------
# Meta `segment_anything` checkpoint naming (sam_vit_{b,l,h}_*.pth): same tensors, different keys

def _meta_attn(prefix: str):
    return {"q": f"{prefix}.q_proj", "k": f"{prefix}.k_proj", "v": f"{prefix}.v_proj", "o": f"{prefix}.out_proj"}


def meta_name_map(cfg: SamConfig) -> List[Tuple[str, str]]:
    """(our name, key in Meta's `Sam.state_dict()`) for tensors that map one-to-one."""
    m = [("enc.patch.b", "image_encoder.patch_embed.proj.bias"),
         ("enc.neck.ln1.w", "image_encoder.neck.1.weight"), ("enc.neck.ln1.b", "image_encoder.neck.1.bias"),
         ("enc.neck.conv2.w", "image_encoder.neck.2.weight"),
         ("enc.neck.ln2.w", "image_encoder.neck.3.weight"), ("enc.neck.ln2.b", "image_encoder.neck.3.bias"),
         ("pe.gauss", "prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"),
         ("dec.mask_tokens", "mask_decoder.mask_tokens.weight"),
         ("dec.ln_final.w", "mask_decoder.transformer.norm_final_attn.weight"),
         ("dec.ln_final.b", "mask_decoder.transformer.norm_final_attn.bias"),
         ("dec.up1.w", "mask_decoder.output_upscaling.0.weight"), ("dec.up1.b", "mask_decoder.output_upscaling.0.bias"),
         ("dec.up_ln.w", "mask_decoder.output_upscaling.1.weight"), ("dec.up_ln.b", "mask_decoder.output_upscaling.1.bias"),
         ("dec.up2.w", "mask_decoder.output_upscaling.3.weight"), ("dec.up2.b", "mask_decoder.output_upscaling.3.bias")]
    for i in range(cfg.depth):
        a, b = f"enc.L{i}", f"image_encoder.blocks.{i}"
        m += [(f"{a}.ln1.w", f"{b}.norm1.weight"), (f"{a}.ln1.b", f"{b}.norm1.bias"),
              (f"{a}.qkv.w", f"{b}.attn.qkv.weight"), (f"{a}.qkv.b", f"{b}.attn.qkv.bias"),
              (f"{a}.rel_h", f"{b}.attn.rel_pos_h"), (f"{a}.rel_w", f"{b}.attn.rel_pos_w"),
              (f"{a}.proj.w", f"{b}.attn.proj.weight"), (f"{a}.proj.b", f"{b}.attn.proj.bias"),
              (f"{a}.ln2.w", f"{b}.norm2.weight"), (f"{a}.ln2.b", f"{b}.norm2.bias"),
              (f"{a}.fc1.w", f"{b}.mlp.lin1.weight"), (f"{a}.fc1.b", f"{b}.mlp.lin1.bias"),
              (f"{a}.fc2.w", f"{b}.mlp.lin2.weight"), (f"{a}.fc2.b", f"{b}.mlp.lin2.bias")]

    def attn(ours, theirs):
        r = []
        for p, q in _meta_attn(theirs).items():
            r += [(f"{ours}.{p}.w", f"{q}.weight"), (f"{ours}.{p}.b", f"{q}.bias")]
        return r
    for i in range(DEC_DEPTH):
        a, b = f"dec.L{i}", f"mask_decoder.transformer.layers.{i}"
        m += attn(f"{a}.self", f"{b}.self_attn")
        m += attn(f"{a}.t2i", f"{b}.cross_attn_token_to_image")
        m += attn(f"{a}.i2t", f"{b}.cross_attn_image_to_token")
        for k in (1, 2, 3, 4):
            m += [(f"{a}.ln{k}.w", f"{b}.norm{k}.weight"), (f"{a}.ln{k}.b", f"{b}.norm{k}.bias")]
        m += [(f"{a}.mlp.fc1.w", f"{b}.mlp.lin1.weight"), (f"{a}.mlp.fc1.b", f"{b}.mlp.lin1.bias"),
              (f"{a}.mlp.fc2.w", f"{b}.mlp.lin2.weight"), (f"{a}.mlp.fc2.b", f"{b}.mlp.lin2.bias")]
    m += attn("dec.final", "mask_decoder.transformer.final_attn_token_to_image")
    for t in range(NUM_MASK_TOKENS):
        for j in range(3):
            m += [(f"dec.hyper{t}.{j}.w", f"mask_decoder.output_hypernetworks_mlps.{t}.layers.{j}.weight"),
                  (f"dec.hyper{t}.{j}.b", f"mask_decoder.output_hypernetworks_mlps.{t}.layers.{j}.bias")]
    for j in range(3):
        m += [(f"dec.iou.{j}.w", f"mask_decoder.iou_prediction_head.layers.{j}.weight"),
              (f"dec.iou.{j}.b", f"mask_decoder.iou_prediction_head.layers.{j}.bias")]
    return m


def to_meta_state_dict(cfg: SamConfig, params: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    d, g = cfg.embed_dim, cfg.grid
    sd = {theirs: params[ours] for ours, theirs in meta_name_map(cfg)}
    sd["image_encoder.patch_embed.proj.weight"] = params["enc.patch.w"].reshape(d, 3, cfg.patch_size, cfg.patch_size)
    sd["image_encoder.pos_embed"] = params["enc.pos"].reshape(1, g, g, d)
    sd["image_encoder.neck.0.weight"] = params["enc.neck.conv1.w"].reshape(cfg.out_chans, d, 1, 1)
    for i in range(4):
        sd[f"prompt_encoder.point_embeddings.{i}.weight"] = params["pe.point"][i:i + 1]
    sd["prompt_encoder.not_a_point_embed.weight"] = params["pe.not_a_point"].reshape(1, -1)
    sd["prompt_encoder.no_mask_embed.weight"] = params["pe.no_mask"].reshape(1, -1)
    sd["mask_decoder.iou_token.weight"] = params["dec.iou_token"].reshape(1, -1)
    return sd


def from_meta_state_dict(cfg: SamConfig, sd) -> Dict[str, np.ndarray]:
    """Meta checkpoint (`torch.load('sam_vit_b_01ec64.pth')`) -> our tensors.  Unused tensors of the
    checkpoint (mask_downscaling.*: the mask-input branch, never taken here) are ignored."""
    def a(x):
        return np.ascontiguousarray(x.detach().cpu().numpy() if hasattr(x, "detach") else x, dtype=np.float32)
    d = cfg.embed_dim
    missing = [theirs for _, theirs in meta_name_map(cfg) if theirs not in sd]
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} tensors of {cfg.name} (wrong variant?), e.g. {missing[0]}")
    p = {ours: a(sd[theirs]) for ours, theirs in meta_name_map(cfg)}
    p["enc.patch.w"] = a(sd["image_encoder.patch_embed.proj.weight"]).reshape(d, -1)
    p["enc.pos"] = a(sd["image_encoder.pos_embed"]).reshape(-1, d)
    p["enc.neck.conv1.w"] = a(sd["image_encoder.neck.0.weight"]).reshape(cfg.out_chans, d)
    p["pe.point"] = np.concatenate([a(sd[f"prompt_encoder.point_embeddings.{i}.weight"]) for i in range(4)], 0)
    p["pe.not_a_point"] = a(sd["prompt_encoder.not_a_point_embed.weight"]).reshape(-1)
    p["pe.no_mask"] = a(sd["prompt_encoder.no_mask_embed.weight"]).reshape(-1)
    p["dec.iou_token"] = a(sd["mask_decoder.iou_token.weight"]).reshape(-1)
    for n, s, _ in param_specs(cfg):
        if tuple(p[n].shape) != tuple(s):
            raise ValueError(f"{n}: checkpoint shape {p[n].shape} != {s} (wrong variant?)")
    return p
