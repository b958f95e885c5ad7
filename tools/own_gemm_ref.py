import sys; sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api
for M, N, K, name in [(4096, 2304, 768, "qkv"), (4096, 3072, 768, "fc1"), (4096, 768, 768, "proj"), (4096, 768, 3072, "fc2"),
                      (32768, 2304, 768, "qkv b8"), (32768, 3072, 768, "fc1 b8"), (4096, 4096, 4096, "4k"), (8192, 8192, 8192, "8k")]:
    ms = api.ext.bench_gemm(M, N, K, 0, iters=50)
    print(f"{name:8s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
