"""Kernel time of the attention kernels alone (dlimg_amd_bench_attention): python tools/attn_variants.py [global|window] [heads] [hd] [batch]
With DLIMGEDIT_TUNING_LIB=1 and DLIMGEDIT_ATTN_ABLATE=n the tuning build's ablated variants of the global kernel are timed (wrong results)."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

kind = sys.argv[1] if len(sys.argv) > 1 else "global"
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hd = int(sys.argv[3]) if len(sys.argv) > 3 else 64
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ms = [api.ext.bench_attention(kind == "global", heads, hd, batch, iters=100) for _ in range(3)]
print(f"{kind} heads {heads} hd {hd} batch {batch} var {os.environ.get('DLIMGEDIT_ATTN_ABLATE', '-')}: "
      + " ".join(f"{m * 1e3:7.1f}" for m in ms) + " us per launch", flush=True)

if kind == "global" and os.environ.get("DLIMGEDIT_ATTN_CHECK"):
    import numpy as np
    rng = np.random.default_rng(0)
    D = heads * hd
    qkv = (rng.standard_normal((4096, 3 * D)) * 1.5).astype(np.float16)
    rel_h = (rng.standard_normal((127, hd)) * 0.3).astype(np.float32)
    rel_w = (rng.standard_normal((127, hd)) * 0.3).astype(np.float32)
    out = api.ext.test_attention(True, qkv, None, rel_h, rel_w, 1, heads, hd).astype(np.float32)
    ref = Path(f"/tmp/attn_var_ref_{heads}_{hd}_{os.getpid() // 100000}.npy")
    if ref.exists():
        r = np.load(ref)
        print("   max |diff| vs the first variant run:", float(np.abs(out - r).max()), "max |out|", float(np.abs(r).max()))
    else:
        np.save(ref, out)
