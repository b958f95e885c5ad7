"""In-kernel clocks of the ping-pong GEMM (tile 9) on the shapes of a TWO-image pass (M = 8192): cycles per K tile, the
clock the chip holds, prologue and epilogue cycles per workgroup -- alone and as four concurrent copies.
DLIMGEDIT_TUNING_LIB=1 python tools/gemm_clock2.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for name, N, K, act, fl in [("qkv", 2304, 768, 0, 1), ("fc1", 3072, 768, 1, 1), ("proj", 768, 768, 0, 3), ("fc2", 768, 3072, 0, 3)]:
    for streams in (1, 4):
        ms, st = api.ext.bench_gemm_stamps(M, N, K, act, iters=20, flavour=fl, tile=9, streams=streams)
        g = (M // 256) * (N // 256)
        nk = K // 64
        raw = st[:g]
        f = raw.astype(np.float64)
        loop_cyc, loop_tk, all_tk = np.median(f[:, 0]), np.median(f[:, 1]), np.median(f[:, 3])
        pro = float(np.median(raw[:, 2] >> np.uint64(32)))
        epi = float(np.median(raw[:, 2] & np.uint64(0xffffffff)))
        ghz = loop_cyc / max(loop_tk, 1) * 0.1
        gf = 2.0 * M * N * K / 1e9
        print(f"{name:5s} M={M} {g:4d} WG x{streams}: {ms * 1e3:7.1f} us/GEMM {gf / ms:6.0f} TF | loop {loop_cyc / nk:6.0f} cyc/Ktile "
              f"{loop_tk * 10 / 1e3:6.2f} us @ {ghz:4.2f} GHz | workgroup {all_tk * 10 / 1e3:6.2f} us | prologue {pro:6.0f} cyc "
              f"({pro / ghz / 1e3:5.2f} us) epilogue {epi:6.0f} cyc ({epi / ghz / 1e3:5.2f} us)", flush=True)
