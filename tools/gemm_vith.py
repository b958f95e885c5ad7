import sys; sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import os
from dlimgedit_amd import api
for M,N,K,name,fl in [(4096,1280,1280,"proj_h",3),(4096,1280,5120,"fc2_h",3),(4096,3840,1280,"qkv_h",1),(4096,5120,1280,"fc1_h",1)]:
    ms = api.ext.bench_gemm(M,N,K,0,iters=30,flavour=fl)
    print(f"  {name:7s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
