"""Times the global attention kernel alone (kernel-trace free: wall clock over repeated calls is dominated by copies, so
this uses the test hook under rocprofv3 --kernel-trace --stats, or prints parity vs the other variant)."""
import os, sys, subprocess
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api
rng = np.random.default_rng(0)
heads, hd = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 64)
D = heads * hd
qkv = (rng.standard_normal((4096, 3 * D)) * 0.5).astype(np.float16)
rel_h = (rng.standard_normal((127, hd)) * 0.1).astype(np.float32)
rel_w = (rng.standard_normal((127, hd)) * 0.1).astype(np.float32)
out = api.ext.test_attention(True, qkv, None, rel_h, rel_w, 1, heads, hd)
np.save(f"/tmp/attn_{os.environ.get('DLIMGEDIT_ATTN_PP', '1')}.npy", out)
print("variant", os.environ.get("DLIMGEDIT_ATTN_PP", "1"), "mean |out|", float(np.abs(out.astype(np.float32)).mean()))
other = f"/tmp/attn_{'0' if os.environ.get('DLIMGEDIT_ATTN_PP', '1') != '0' else '1'}.npy"
if os.path.exists(other):
    o = np.load(other)
    print("max |diff| vs other variant", float(np.abs(out.astype(np.float32) - o.astype(np.float32)).max()))
