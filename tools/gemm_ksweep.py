"""Fixed cost vs per-K cost of a GEMM tile configuration: time at K = 768, 1536, 3072 (same M, N) -> intercept (launch +
prologue + epilogue) and slope (main loop per 64 of K).  python tools/gemm_ksweep.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

Ks = [768, 1536, 3072]
TILES = [int(t) for t in sys.argv[1:]] or [7, 2]
for name, N, act, fl in [("qkv-like N=2304 plain f16 out", 2304, 0, 0), ("N=2304 LN-folded + bias", 2304, 0, 1),
                         ("fc1-like N=3072 LN + GELU", 3072, 1, 1), ("fc1-like N=3072 LN, no GELU", 3072, 0, 1),
                         ("proj-like N=768 resid+stats", 768, 0, 3), ("N=768 plain", 768, 0, 0)]:
    for tile in TILES:
        for streams in (1, 4):
            ts = [api.ext.bench_gemm(4096, N, K, act, iters=30, flavour=fl, tile=tile, streams=streams) * 1e3 for K in Ks]
            slope, icpt = np.polyfit(Ks, ts, 1)
            print(f"{name:32s} tile {tile} x{streams}: " + "  ".join(f"K={K}: {t:6.1f} us" for K, t in zip(Ks, ts)) +
                  f"  | fixed {icpt:5.1f} us, {slope * 64:5.2f} us per 64 of K", flush=True)
