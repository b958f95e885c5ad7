import os, subprocess, sys
code = r'''
import sys; sys.path.insert(0, %r)
from dlimgedit_amd import api
import os
if os.environ.get('TOOL_GEMM_TILE'): api.ext.force_gemm_tile(int(os.environ['TOOL_GEMM_TILE']))
for M,N,K,name in [(4096,768,3072,"fc2"),(4096,768,768,"proj"),(4096,2304,768,"qkv"),(4096,1280,5120,"fc2_h")]:
    ms = api.ext.bench_gemm(M,N,K,0,iters=30)
    print(f"  {name:5s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
''' % str(__import__("pathlib").Path(__file__).resolve().parent.parent)
for tile in ["2", "0"]:
    for abl in ["0", "1", "2"]:
        env = dict(os.environ, TOOL_GEMM_TILE=tile, DLIMGEDIT_GEMM_ABLATE=abl)
        print(f"tile={tile} ablate={abl} (0 real, 1 streaming only, 2 MFMA+LDS only)", flush=True)
        subprocess.run([sys.executable, "-c", code], env=env)
for z in ["1"]:
    env = dict(os.environ, TOOL_GEMM_TILE="2", DLIMGEDIT_BENCH_ZERO="1")
    print("tile=2 zero operands", flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
