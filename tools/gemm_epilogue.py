"""Prologue / K loop / epilogue of the ping-pong GEMM flavours in cycles (in-kernel stamps, tuning build), four images per launch:

    python -m dlimgedit_amd.build --tuning && gpurun -- 'DLIMGEDIT_TUNING_LIB=1 python tools/gemm_epilogue.py'

Ablations of the stream writers' epilogue are compile-time (WRONG results, tuning build only):
    DLIMG_TUNING_DEFS="-DDLIMG_NO_RESID_READ" | "-DDLIMG_NO_STATS_MATH" | "-DDLIMG_NO_STREAM_STORE"  python -m dlimgedit_amd.build --tuning
(copy lib/libdlimgedit_tuning.so to lib/libdlimgedit_<name>.so per variant and pass DLIMGEDIT_TUNING_LIB=libdlimgedit_<name>.so)."""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from dlimgedit_amd import api

M = int(os.environ.get("ROWS", "16384"))
TILE = int(os.environ.get("TILE", "9"))       # 9: 256 x 256, 10: 128 x 256, 11: 64 x 256
BM = {9: 256, 10: 128, 11: 64}[TILE]
ROWS = [("qkv plain", 2304, 768, 0, 0), ("qkv LN", 2304, 768, 0, 1), ("fc1 LN+gelu", 3072, 768, 1, 1),
        ("proj fp32 in place", 768, 768, 0, 2), ("proj fp32+copy+stats", 768, 768, 0, 3),
        ("proj pair", 768, 768, 0, 6), ("proj pair+stats", 768, 768, 0, 5), ("fc2 pair+stats", 768, 3072, 0, 5)]
only = sys.argv[1:]
for name, N, K, act, fl in ROWS:
    if only and not any(o in name for o in only):
        continue
    ms, st = api.ext.bench_gemm_stamps(M, N, K, act, iters=20, flavour=fl, tile=TILE, streams=1)
    g = (M // BM) * (N // 256)
    raw = st[:g]
    f = raw.astype(np.float64)
    loop_cyc, loop_tk, all_tk = np.median(f[:, 0]), np.median(f[:, 1]), np.median(f[:, 3])
    pro = float(np.median(raw[:, 2] >> np.uint64(32)))       # (the 128- and 64-row kernel stamps loop and epilogue only)
    epi = float(np.median(raw[:, 2] & np.uint64(0xffffffff)))
    ghz = loop_cyc / max(loop_tk, 1) * 0.1
    print(f"{name:22s}: {ms*1e3:7.1f} us | loop {loop_cyc:7.0f} cyc | prologue {pro:6.0f} | epilogue {epi:6.0f} | wg {all_tk*10/1e3:6.2f} us @ {ghz:4.2f} GHz", flush=True)
