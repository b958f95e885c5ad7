"""Model geometry of the Segment-Anything variants served behind Segmentation::process().

The reference hard-codes one encoder graph by file name (/root/reference/src/segmentation.cpp:14-24)
and reads its shapes back from the session (segmentation.cpp:35-41).  Here the same information is a
small table: the HIP executor and the oracle are both driven by it.  Spatial geometry (1024 px input,
16 px patches, 64x64 token grid, 14x14 attention windows, 256-channel embedding) is fixed by the
reference's `image_input_size = 1024` (segmentation.cpp:17) and its 1x256x64x64 embedding contract;
only width/depth/heads vary between variants.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple

IMAGE_SIZE = 1024          # /root/reference/src/segmentation.cpp:17
PATCH = 16
GRID = IMAGE_SIZE // PATCH  # 64
TOKENS = GRID * GRID        # 4096
WINDOW = 14
EMBED_OUT = 256             # channels of the cached image embedding (1x256x64x64)

# SAM pixel statistics applied inside the encoder graph (export_models.py:26 use_preprocess=True)
PIXEL_MEAN = (123.675, 116.28, 103.53)
PIXEL_STD = (58.395, 57.12, 57.375)

# mask decoder (identical for every encoder variant)
DEC_DIM = 256
DEC_HEADS = 8
DEC_MLP = 2048
DEC_DEPTH = 2
DEC_DOWNSAMPLE = 2
NUM_MASK_TOKENS = 4
IOU_HIDDEN = 256
LOWRES = 256                # low-res mask logits are 256x256


@dataclass(frozen=True)
class SamConfig:
    name: str
    embed_dim: int
    depth: int
    num_heads: int
    global_attn_indexes: Tuple[int, ...]
    mlp_ratio: int = 4
    image_size: int = IMAGE_SIZE
    patch_size: int = PATCH
    window_size: int = WINDOW
    out_chans: int = EMBED_OUT

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def mlp_dim(self) -> int:
        return self.embed_dim * self.mlp_ratio

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    def encoder_flops(self) -> float:
        """Algorithmic FLOPs of one image encode (2 per MAC), SURVEY.md §8(d) accounting:
        linears on the real tokens, windowed attention on padded windows, rel-pos einsums."""
        n = self.grid * self.grid
        d, hd, h = self.embed_dim, self.head_dim, self.num_heads
        lin = self.depth * 2 * n * d * (3 * d + d + 2 * self.mlp_dim)
        n_glob = len(self.global_attn_indexes)
        n_win = self.depth - n_glob
        g = self.grid
        glob = n_glob * (2 * 2 * n * n * d + 2 * 2 * n * g * hd * h)
        w = self.window_size
        nw = ((g + w - 1) // w) ** 2
        win = n_win * (2 * 2 * nw * (w * w) ** 2 * d + 2 * 2 * nw * (w * w) * w * hd * h)
        patch = 2 * n * d * 3 * self.patch_size ** 2
        neck = 2 * n * self.out_chans * d + 2 * n * self.out_chans * self.out_chans * 9
        return float(lin + glob + win + patch + neck)


CONFIGS = {
    "vit_b": SamConfig("vit_b", 768, 12, 12, (2, 5, 8, 11)),
    "vit_l": SamConfig("vit_l", 1024, 24, 16, (5, 11, 17, 23)),
    "vit_h": SamConfig("vit_h", 1280, 32, 16, (7, 15, 23, 31)),
    # reduced-width/depth variant with the same spatial geometry: used by the parity tests so the
    # CPU oracle finishes in well under a second while every kernel family is exercised
    # (one windowed layer with padding, one global layer, neck, full decoder).
    "vit_test": SamConfig("vit_test", 128, 2, 2, (1,)),
    # head_dim 80 coverage (ViT-H's head size) at test cost
    "vit_test80": SamConfig("vit_test80", 320, 2, 4, (1,)),
}


def get_config(name: str) -> SamConfig:
    try:
        return CONFIGS[name]
    except KeyError:
        raise ValueError(f"unknown SAM variant '{name}' (known: {sorted(CONFIGS)})") from None
