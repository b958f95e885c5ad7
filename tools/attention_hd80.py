import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api
rng = np.random.default_rng(0)
heads, hd = 16, 80
D = heads * hd
qkv = rng.standard_normal((4096, 3 * D)).astype(np.float16)
bias = (0.5 * rng.standard_normal(3 * D)).astype(np.float32)
for is_global, span in ((False, 14), (True, 64)):
    rel_h = (0.1 * rng.standard_normal((2 * span - 1, hd))).astype(np.float32)
    rel_w = (0.1 * rng.standard_normal((2 * span - 1, hd))).astype(np.float32)
    for _ in range(3):
        api.ext.test_attention(is_global, qkv, bias, rel_h, rel_w, 1, heads, hd)
