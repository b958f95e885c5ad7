"""Runs the global (or windowed) attention kernel alone a few times: target of rocprofv3 --pmc / --kernel-trace runs.
python tools/attn_probe.py [global|window] [heads] [hd] [reps]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

kind = sys.argv[1] if len(sys.argv) > 1 else "global"
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
hd = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rng = np.random.default_rng(0)
D = heads * hd
qkv = (rng.standard_normal((4096, 3 * D)) * 0.5).astype(np.float16)
span = 64 if kind == "global" else 14
rel_h = (rng.standard_normal((2 * span - 1, hd)) * 0.1).astype(np.float32)
rel_w = (rng.standard_normal((2 * span - 1, hd)) * 0.1).astype(np.float32)
bias = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
for _ in range(reps):
    out = api.ext.test_attention(kind == "global", qkv, bias, rel_h, rel_w, 1, heads, hd)
print("ok", float(np.abs(out.astype(np.float32)).mean()))

import os
if kind == "global" and int(os.environ.get("DLIMGEDIT_ATTN_ABLATE", "0")) & 4:
    # cycle stamps of the tuning build (DLIMGEDIT_TUNING_LIB=1 DLIMGEDIT_ATTN_ABLATE=4, +8 / +16 / +1 for the ablations):
    # workgroups with blockIdx % 3 == 0 overwrite the first rows of their own output
    o = np.ascontiguousarray(out)                      # [4096][D] f16
    rows = []
    nwg = heads * 16

    def xcd_remap(b):                                  # as device_common.hpp (the kernel maps blockIdx -> logical block)
        q, r = nwg // 8, nwg % 8
        xcd, k = b % 8, b // 8
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + k

    for b in range(0, nwg, 3):
        bid = xcd_remap(b)
        qblk, head = bid % 16, bid // 16
        r0 = o[qblk * 256, head * hd:head * hd + 32].view(np.uint64).astype(np.float64)      # A: tm tmb tx txb, B: ...
        r1 = o[qblk * 256 + 1, head * hd:head * hd + 16].view(np.uint64).astype(np.float64)
        r2 = o[qblk * 256 + 2, head * hd:head * hd + 16].view(np.uint64).astype(np.float64)
        rows.append(np.concatenate([r0, r1, r2]))
        if len(rows) <= 2:
            w = np.concatenate([o[qblk * 256 + 3, head * hd:head * hd + 32].view(np.uint64), o[qblk * 256 + 4, head * hd:head * hd + 32].view(np.uint64)]).astype(np.int64)
            print("   wg", b, "per wave: start", [int(w[2 * i] - w[0]) for i in range(8)], "reaches the last barrier of the prologue", [int(w[2 * i + 1] - w[0]) for i in range(8)])
    raw = np.array(rows)
    print("prologue pieces (requests | relw | relh | K/V to LDS + barrier):", [int(np.median(raw[:, 12 + i])) for i in range(4)], "raises of the reference maximum", int(np.median(raw[:, 11])))
    cyc, ticks, pro = np.median(raw[:, 8]), np.median(raw[:, 9]), np.median(raw[:, 10])
    print(f"loop: {cyc:.0f} cycles = {cyc / 64:.0f} per tile, {ticks * 0.01:.1f} us, clock {cyc / ticks * 0.1:.2f} GHz; prologue {pro:.0f} cycles")
    for g, name in ((0, "group A"), (1, "group B")):
        tm, tmb, tx, txb = (np.median(raw[:, g * 4 + i]) / 21.0 for i in range(4))     # stamped: one of three unrolled tiles
        print(f"{name}: per tile  M work {tm:6.0f} cyc + barrier wait {tmb:6.0f} | X work {tx:6.0f} cyc + barrier wait {txb:6.0f}  (sum {tm+tmb+tx+txb:6.0f})")
