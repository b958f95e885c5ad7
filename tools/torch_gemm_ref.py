"""Reference point only (not part of the product): what the vendor GEMM (hipBLASLt through torch.matmul) reaches on
the encoder shapes, to judge how far the hand-written kernels are from the practical ceiling of this box."""
import torch

def bench(M, N, K, iters=50):
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
    ms = e0.elapsed_time(e1) / iters
    return ms

for M, N, K, name in [(4096, 2304, 768, "qkv"), (4096, 3072, 768, "fc1"), (4096, 768, 768, "proj"), (4096, 768, 3072, "fc2"),
                      (32768, 2304, 768, "qkv b8"), (32768, 3072, 768, "fc1 b8"), (4096, 4096, 4096, "4k"), (8192, 8192, 8192, "8k")]:
    ms = bench(M, N, K)
    print(f"{name:8s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
