"""Runs one GEMM shape a few times (for rocprofv3 --pmc): python tools/gemm_one.py M N K [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api
M, N, K = map(int, sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ms = api.ext.bench_gemm(M, N, K, 0, iters=iters)
print(f"{M}x{N}x{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF")
