"""Ablation of the GEMM kernel on the GPU box: real vs streaming-only vs MFMA-only, per tile config."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys; sys.path.insert(0, %r)
from dlimgedit_amd import api
for M,N,K,name in [(4096,2304,768,"qkv"),(4096,3072,768,"fc1"),(4096,768,768,"proj"),(4096,768,3072,"fc2"),(4096,4096,4096,"4k")]:
    try:
        ms = api.ext.bench_gemm(M,N,K,0,iters=20)
        print(f"  {name:5s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
    except Exception as e:
        print("  ", name, "n/a", str(e)[:60])
''' % str(ROOT)
for tile in ["2", "8", "6", "7"]:
    for abl in ["0"]:
        env = dict(os.environ, DLIMGEDIT_GEMM_TILE=tile, DLIMGEDIT_GEMM_ABLATE=abl)
        print(f"tile={tile} ablate={abl}", flush=True)
        subprocess.run([sys.executable, "-c", code], env=env)
