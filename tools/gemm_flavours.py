"""Cost of the GEMM epilogue flavours on the encoder shapes, one process, same box: plain vs LayerNorm-folded
consumer (qkv / fc1) and residual writer vs residual writer + f16 copy + row statistics (proj / fc2)."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys; sys.path.insert(0, %r)
from dlimgedit_amd import api
import os
if os.environ.get('TOOL_GEMM_TILE'): api.ext.force_gemm_tile(int(os.environ['TOOL_GEMM_TILE']))
for rep in range(2):
    print("  plain, no bias: qkv {:.1f} us  fc1 {:.1f} us".format(api.ext.bench_gemm(4096,2304,768,0,iters=50)*1e3, api.ext.bench_gemm(4096,3072,768,0,iters=50)*1e3))
    for M,N,K,name,fl in [(4096,2304,768,"qkv",(4,1)),(4096,3072,768,"fc1",(4,1)),(4096,768,768,"proj",(2,3)),(4096,768,3072,"fc2",(2,3))]:
        t = [api.ext.bench_gemm(M,N,K,0,iters=50,flavour=f)*1e3 for f in fl]
        print(f"  {name:5s} flavour {fl[0]}: {t[0]:7.1f} us   flavour {fl[1]}: {t[1]:7.1f} us   delta {t[1]-t[0]:+6.1f} us", flush=True)
''' % str(ROOT)
for tile in sys.argv[1:] or ["", "7", "4"]:
    env = dict(os.environ)
    if tile:
        env["TOOL_GEMM_TILE"] = tile
    print(f"tile={tile or 'auto'}", flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
