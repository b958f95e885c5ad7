"""GEMM kernel sweep on the GPU box: TFLOP/s per shape (device-resident random operands)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

shapes = [
    (4096, 2304, 768, 0, "qkv b1"), (4096, 768, 768, 0, "proj b1"), (4096, 3072, 768, 1, "fc1 b1 gelu"),
    (4096, 3072, 768, 0, "fc1 b1 nogelu"), (4096, 768, 3072, 0, "fc2 b1"),
    (32768, 2304, 768, 0, "qkv b8"), (32768, 768, 768, 0, "proj b8"), (32768, 3072, 768, 1, "fc1 b8 gelu"),
    (32768, 768, 3072, 0, "fc2 b8"),
    (4096, 3840, 1280, 0, "qkv vit_h"), (4096, 1280, 5120, 0, "fc2 vit_h"),
    (4096, 4096, 4096, 0, "4k cube"), (8192, 8192, 4096, 0, "8k"),
]
for M, N, K, act, name in shapes:
    ms = api.ext.bench_gemm(M, N, K, act, iters=20)
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
