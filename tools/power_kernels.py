"""Package power, shader clock and energy per image, kernel by kernel.

Each of the encoder's dominant kernels is looped ALONE for `seconds` (default 3) on device-resident random operands -- the
launches of a four-image pass at ViT-B, through the kernels-alone hooks of lib/libdlimgedit_test.so -- while a host thread
samples the package power and the shader clock; then the official bench command runs as one long block with the same
sampler beside it (the steady state of the lanes).  Prints one table:

    kernel | us per launch | TFLOP/s | W (median) | sclk MHz (median) | launches per image | mJ per image

so that J per image of the steady state can be set against the sum over the kernels, and "the package sits at its limit"
 can be reproduced from profiles/ like every other claim.

    gpurun -- 'python3 tools/power_kernels.py [vit_b|vit_h] [seconds] > gpurun_out/power_kernels.txt'

Power and clock come from rocm-smi (--showpower --showclocks), polled about three times a second; the numbers are the
board's own telemetry, not an in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6: sclk reads up to 10 % above it).
"""
import re
import statistics
import subprocess
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dlimgedit_amd import api                      # noqa: E402
from dlimgedit_amd.sam_config import get_config    # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
cfg = get_config(model)
D, F, H, hd = cfg.embed_dim, cfg.mlp_dim, cfg.num_heads, cfg.head_dim
IMAGES = 4 if D <= 1024 else 3                     # images per pass of the step queue (ext_api.cpp, step_queue_width)
M = IMAGES * 4096
n_glob = len(cfg.global_attn_indexes)
n_win = cfg.depth - n_glob


class Sampler:
    """rocm-smi polled in a thread between start() and stop(); medians of what it saw."""

    def __init__(self):
        self.power, self.sclk, self._stop, self._t = [], [], threading.Event(), None

    def _poll(self):
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                out = ""
            p = re.search(r"Power \(W\):\s*([\d.]+)", out)
            c = re.search(r"sclk clock level:\s*\d+:?\s*\(?([\d.]+)\s*Mhz", out, re.I)
            if p:
                self.power.append(float(p.group(1)))
            if c:
                self.sclk.append(float(c.group(1)))
            self._stop.wait(0.15)

    def start(self):
        self.power, self.sclk = [], []
        self._stop.clear()
        self._t = threading.Thread(target=self._poll)
        self._t.start()

    def stop(self):
        self._stop.set()
        self._t.join()
        # the first sample may predate the load, the last may trail it
        pw = self.power[1:-1] if len(self.power) > 4 else self.power
        ck = self.sclk[1:-1] if len(self.sclk) > 4 else self.sclk
        return (statistics.median(pw) if pw else float("nan"), max(pw) if pw else float("nan"),
                statistics.median(ck) if ck else float("nan"), len(pw))


sampler = Sampler()
rows = []


def looped(name, launch_ms, flops, per_image):
    """launch_ms(iters) -> average ms per launch over `iters` back-to-back launches"""
    ms = launch_ms(30)
    iters = max(30, int(seconds / (ms * 1e-3)))
    sampler.start()
    time.sleep(0.3)
    ms = launch_ms(iters)
    w, wmax, mhz, n = sampler.stop()
    mj = w * (ms * 1e-3) / IMAGES * per_image * 1e3
    rows.append((name, ms * 1e3, flops / (ms * 1e-3) / 1e12, w, wmax, mhz, per_image, mj, n))
    print(f"{name:34s} {ms * 1e3:8.1f} us {flops / (ms * 1e-3) / 1e12:7.0f} TF  {w:6.0f} W (max {wmax:4.0f})  {mhz:5.0f} MHz  "
          f"x {per_image:3d} per image = {mj:7.1f} mJ per image   [{n} samples]", flush=True)


ext = api.ext
print(f"# {model}: kernels of a {IMAGES}-image pass looped alone for {seconds:.0f} s each (lib/libdlimgedit_test.so hooks), rocm-smi beside them")
sampler.start(); time.sleep(1.5); idle = sampler.stop()
print(f"idle: {idle[0]:.0f} W, sclk {idle[2]:.0f} MHz")
looped("qkv   (LayerNorm folded, f16 out)", lambda it: ext.bench_gemm(M, 3 * D, D, 0, iters=it, flavour=1, tile=9, shared=True), 2.0 * M * 3 * D * D, cfg.depth)
looped("fc1   (LayerNorm folded, GELU)", lambda it: ext.bench_gemm(M, F, D, 1, iters=it, flavour=1, tile=9, shared=True), 2.0 * M * F * D, cfg.depth)
looped("proj  (stream writer, K = D)", lambda it: ext.bench_gemm(M, D, D, 0, iters=it, flavour=5, tile=9, shared=True), 2.0 * M * D * D, cfg.depth)
looped("fc2   (stream writer, K = 4 D)", lambda it: ext.bench_gemm(M, D, F, 0, iters=it, flavour=5, tile=9, shared=True), 2.0 * M * D * F, cfg.depth)
looped("global attention", lambda it: ext.bench_attention(True, H, hd, IMAGES, iters=it),
       IMAGES * (4.0 * 4096 * 4096 * D + 4.0 * 4096 * 64 * hd * H), n_glob)
looped("windowed attention", lambda it: ext.bench_attention(False, H, hd, IMAGES, iters=it),
       IMAGES * 25.0 * (4.0 * 196 * 196 * D + 4.0 * 196 * 14 * hd * H), n_win)
total_mj = sum(r[7] for r in rows)
print(f"sum over these kernels: {total_mj:.0f} mJ per image, kernels running one at a time "
      f"({sum(r[1] * r[6] for r in rows) / IMAGES / 1e3:.3f} ms per image)")

# ---- the steady state of the lanes: the official command as one long block
import json  # noqa: E402
steps = 3000 if D <= 768 else 600
cmd = [sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--model", model, "--steps", str(steps), "--warmup", "5",
       "--repeats", "3", "--no-cpu-baseline", "--no-abi-path", "--no-config-legs"]
# sampled from start to end: model load and warm-up draw far less than the timed blocks, so the steady state is read from the
# samples within 10 % of the highest power seen (their median, with the clock samples taken at the same moments)
proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
sampler.start()
out, _ = proc.communicate(timeout=900)
sampler._stop.set()
sampler._t.join()
pairs = list(zip(sampler.power, sampler.sclk))
line = [ln for ln in out.splitlines() if ln.startswith("{")]
if line and pairs:
    top = max(p for p, _ in pairs)
    steady = [(p, c) for p, c in pairs if p >= 0.9 * top]
    w, wmax, mhz, n = statistics.median(p for p, _ in steady), top, statistics.median(c for _, c in steady), len(steady)
    d = json.loads(line[-1])
    print(f"steady state ({' '.join(cmd[1:])}): {d['value']:.1f} images/s, {w:.0f} W (max {wmax:.0f}), sclk {mhz:.0f} MHz "
          f"[{n} of {len(pairs)} samples within 10 % of the maximum] = {w / d['value'] * 1e3:.0f} mJ per image; chip_frac {d['roofline']['chip_frac']:.3f}")
else:
    print("steady state: the bench printed no line")
