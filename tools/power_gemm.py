"""Where the watts go (measurement aid, not part of the product; torch is used here only as a way to call hipBLASLt).

For each of the four ViT-B encoder GEMM shapes, FOUR concurrent streams run the same problem back to back for a few
seconds -- once through hipBLASLt (torch.matmul) and once through this library's kernels (dlimg_amd_bench_gemm_streams,
real epilogues) -- while `rocm-smi --showpower --showclocks` is sampled beside them.  Prints TFLOP/s (aggregate over
the four streams), package power and sclk for both, so the two can be compared at the same power cap.

  python tools/power_gemm.py [seconds_per_case]
"""
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402

from dlimgedit_amd import api  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
SHAPES = [("qkv", 4096, 2304, 768, 0, 1), ("fc1", 4096, 3072, 768, 1, 1), ("proj", 4096, 768, 768, 0, 3),
          ("fc2", 4096, 768, 3072, 0, 3), ("4k", 4096, 4096, 4096, 0, 0)]


class Sampler:
    """rocm-smi once every ~0.4 s on a host thread; keeps (power W, sclk MHz) pairs."""

    def __init__(self):
        self.samples = []
        self._stop = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=10).stdout
            except Exception:
                break
            p = re.search(r"Power \(W\):\s*([0-9.]+)", out)
            c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
            if p:
                self.samples.append((float(p.group(1)), int(c.group(1)) if c else 0))
            self._stop.wait(0.4)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._t.join(timeout=15)

    def summary(self):
        if not self.samples:
            return "power n/a"
        s = self.samples[1:] or self.samples          # the first sample straddles the start
        pw = sorted(x[0] for x in s)
        ck = sorted(x[1] for x in s)
        return f"power median {pw[len(pw) // 2]:6.0f} W max {pw[-1]:6.0f} W | sclk median {ck[len(ck) // 2]} MHz ({len(s)} samples)"


def blaslt_streams(M, N, K, seconds, streams=4):
    a = [torch.randn(M, K, device="cuda", dtype=torch.float16) for _ in range(streams)]
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
    out = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for i in range(streams):
        with torch.cuda.stream(ss[i]):
            torch.matmul(a[i], w.t(), out=out[i])
    torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            for i in range(streams):
                with torch.cuda.stream(ss[i]):
                    torch.matmul(a[i], w.t(), out=out[i])
            n += streams
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def ours_streams(M, N, K, act, flavour, seconds, streams=4):
    # one long call (operand set-up on the host would otherwise leave the GPU idle between short calls)
    ms = api.ext.bench_gemm(M, N, K, act, iters=50, flavour=flavour, tile=-1, shared=True, streams=streams)
    iters = max(50, int(seconds / (ms * 1e-3) / streams))
    ms = api.ext.bench_gemm(M, N, K, act, iters=iters, flavour=flavour, tile=-1, shared=True, streams=streams)
    return ms * 1e-3


def main():
    print(f"{SECONDS:.1f} s per case, 4 concurrent streams of the same problem; TF = aggregate over the streams", flush=True)
    for name, M, N, K, act, fl in SHAPES:
        gf = 2.0 * M * N * K / 1e9
        with Sampler() as s:
            sec = blaslt_streams(M, N, K, SECONDS)
        print(f"{name:5s} hipBLASLt   {sec * 1e6:7.1f} us/GEMM {gf / sec / 1e3:7.0f} TF | {s.summary()}", flush=True)
        time.sleep(1.0)
        with Sampler() as s:
            sec = ours_streams(M, N, K, act, fl, SECONDS)
        print(f"{name:5s} this build  {sec * 1e6:7.1f} us/GEMM {gf / sec / 1e3:7.0f} TF | {s.summary()}", flush=True)
        time.sleep(1.0)


if __name__ == "__main__":
    main()
