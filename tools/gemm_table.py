"""The encoder GEMMs with their real epilogues, per tile configuration, alone on the chip (1 stream) and as four
concurrent copies (the regime of the execution lanes); hipBLASLt (torch.matmul, no epilogue) beside them.
python tools/gemm_table.py [vit_b|vit_h] [tiles...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
tiles = [int(t) for t in sys.argv[2:]] or [-1, 0, 1, 2, 3, 6, 7, 8]
D, MLP = (768, 3072) if variant == "vit_b" else (1280, 5120)
shapes = [("qkv", 4096, 3 * D, D, 0, 1), ("proj", 4096, D, D, 0, 3), ("fc1", 4096, MLP, D, 1, 1), ("fc2", 4096, D, MLP, 0, 3)]

try:
    import torch
    def blaslt(M, N, K, iters=50):
        a = torch.randn(M, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
        for _ in range(5):
            torch.matmul(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(a, w.t())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
except Exception:       # pragma: no cover
    blaslt = None

for name, M, N, K, act, fl in shapes:
    gf = 2.0 * M * N * K / 1e9
    line = f"{name:5s} {gf:6.1f} GF"
    if blaslt:
        ms = blaslt(M, N, K)
        line += f" | hipBLASLt {ms * 1e3:6.1f} us {gf / ms:6.0f} TF"
    print(line, flush=True)
    for t in tiles:
        for shared in ((False, True) if t < 0 else (False,)):
            try:
                m1 = api.ext.bench_gemm(M, N, K, act, iters=40, flavour=fl, tile=t, shared=shared, streams=1)
                m4 = api.ext.bench_gemm(M, N, K, act, iters=20, flavour=fl, tile=t, shared=shared, streams=4)
            except api.Error as e:
                continue
            tag = f"tile {t}" if t >= 0 else ("auto/shared" if shared else "auto/alone")
            print(f"      {tag:12s} 1 stream {m1 * 1e3:6.1f} us {gf / m1:6.0f} TF | 4 streams {m4 * 1e3:6.1f} us/GEMM {gf / m4:6.0f} TF",
                  flush=True)
