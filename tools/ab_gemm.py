"""Same-box A/B of two builds of the library on the encoder GEMM shapes (4 concurrent streams, real epilogues):
python tools/ab_gemm.py   -- runs itself twice per round as child processes, product build vs DLIMGEDIT_TUNING_LIB=1 build."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    from dlimgedit_amd import api
    out = []
    for name, M, N, K, act, fl in (("qkv", 4096, 2304, 768, 0, 1), ("fc1", 4096, 3072, 768, 1, 1), ("qkv2", 8192, 2304, 768, 0, 1),
                                   ("fc12", 8192, 3072, 768, 1, 1)):
        ms = min(api.ext.bench_gemm(M, N, K, act, iters=300, flavour=fl, tile=-1, shared=True, streams=4) for _ in range(3))
        out.append(f"{name} {ms * 1e3:.2f}")
    print(" | ".join(out))
else:
    for rnd in range(4):
        for tag, env in (("new", {}), ("old", {"DLIMGEDIT_TUNING_LIB": "1"})):
            r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True)
            print(tag, r.stdout.strip() or r.stderr[-300:], flush=True)
