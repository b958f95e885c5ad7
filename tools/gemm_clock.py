"""In-kernel clock and cycle counts of the ping-pong GEMM (tile 9): shader cycles per K tile of 64 and the clock the chip
holds (cycles / 100 MHz ticks), alone and with four concurrent copies, at 144 and at >= 256 workgroups."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api

for name, M, N, K, act, fl, tile in [("qkv 144 WG", 4096, 2304, 768, 0, 1, 9), ("fc1 192 WG", 4096, 3072, 768, 1, 1, 9),
                                     ("4096^3", 4096, 4096, 4096, 0, 0, 9), ("proj 96 WG t10", 4096, 768, 768, 0, 3, 10),
                                     ("fc2 96 WG t10", 4096, 768, 3072, 0, 3, 10), ("4096^3 t10", 4096, 4096, 4096, 0, 0, 10)]:
    for streams in (1, 4):
        ms, st = api.ext.bench_gemm_stamps(M, N, K, act, iters=20, flavour=fl, tile=tile, streams=streams)
        g = (M // (256 if tile == 9 else 128)) * (N // 256)
        nk = K // 64
        raw = st[:g]
        st = raw.astype(np.float64)
        loop_cyc, loop_tk, _, all_tk = (np.median(st[:, i]) for i in range(4))
        pro_cyc = float(np.median(raw[:, 2] >> np.uint64(32)))
        epi_cyc = float(np.median(raw[:, 2] & np.uint64(0xffffffff)))
        ghz = loop_cyc / max(loop_tk, 1) * 0.1
        print(f"{name:14s} x{streams}: {ms * 1e3:7.1f} us/GEMM | main loop {loop_cyc / nk:7.0f} cyc per K tile (ideal 2048 / 1024), "
              f"{loop_tk * 10 / 1e3:6.2f} us, clock {ghz:4.2f} GHz | whole kernel {all_tk * 10 / 1e3:6.2f} us "
              f"(prologue {pro_cyc:6.0f} cyc, epilogue {epi_cyc:6.0f} cyc)", flush=True)
