#!/usr/bin/env python3
"""Throughput benchmark of the Segment-Anything hot path (BASELINE.json metric:
images/sec encode+mask @1024x1024).

  python bench.py --gpus N --steps K --warmup W [--model vit_b] [--batch B]

One process per GPU (N > 1: launched by torch.distributed.run, RCCL only for the barrier and the
max-over-ranks of the elapsed time: images are independent, there is no data-path collective).
A step = one pass of the hot path over `batch` synthetic 1024x1024 RGBA images that are already
resident in HBM: pre-process -> ViT encoder -> prompt encoder + mask decoder (one point prompt per
image, single-mask mode) -> bilinear upsample + threshold, masks left in HBM.  At N = 1 the default
workload is BASELINE.json configs[1] (ViT-B, batch 1, one point prompt).

Timing: the block of K steps is timed `--repeats` times (default 21), each repeat bracketed by a barrier and
torch.cuda.synchronize() on both sides and reduced to the MAX over ranks; `value` comes from the MEDIAN repeat
(`ms_per_step` likewise; every repeat's time is in `repeat_ms`).

Besides the contract fields the JSON line carries
  roofline      dominant kernel (the f16 MFMA GEMMs of the encoder): algorithmic FLOPs / the kernels' own execution
                time (HIP events attached to each dispatch) in a profiled repeat that runs on ONE lane, i.e. every
                kernel alone on the chip; chip_frac = all FLOPs of a step / ms_per_step / peak, the only figure that
                sees the lanes overlap
  abi_path      images/s through the drop-in ABI itself (host pixels in, host masks out, PCIe and host copies
                included): slots 3-4 from one and from several host threads, slots 13-14 with 8 images per call
  cpu_baseline  the CPU oracle (oracle/sam_oracle.py, a port: the reference's onnxruntime path cannot
                be built here) timed on this host on one image of the same workload
  encoder_only  images/s of pre-processing + encoder alone; decode_only: prompts/s on a cached embedding
  mask_iou      IoU of the HIP mask against the oracle's mask for that image (+ logits_check: max-abs error of the
                decoder's logits / IoU predictions / the embedding against the oracle, fraction of near-zero logits)
  stages        per-stage breakdown of the profiled repeat (profile_mode says how it was taken)
  rccl          N > 1: ranks counted by an all-reduce of ones, and the optional gather of the finished masks to
                every rank (RCCL all_gather over xGMI, after the timed region: no collective is on the data path)
  inproc_replicas  N > 1: the same node driven the way a host of the reference would drive it -- ONE process, ONE
                Environment over all N GPUs (DLIMGEDIT_DEVICES), 8 images per GPU per call through slots 13 / 14 of the
                drop-in table, masks gathered into one buffer on GPU 0 by the library (peer copies over xGMI)

Launching: `python bench.py --gpus N` from a bare shell starts its own N ranks (a parent that never touches the GPU runs
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and relays its output); under a launcher that
has already set WORLD_SIZE the ranks run directly.  `--stub-device` replaces the device by a host stand-in so the N > 1
orchestration (barriers, max over ranks, rank count, gather, the line's assembly) runs on CPU (tests/test_bench_cpu.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

# Hardware queues of the HIP runtime: four by default, shared by every stream of the process; the library's execution
# lanes each want their own.  Read once when the runtime initialises, so it is set before anything can touch the GPU
# (torch may initialise HIP before the library is loaded; the library itself asks for eight queues at load otherwise).
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    os.environ.setdefault("DLIMGEDIT_PLAIN_STREAMS", "1")

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_F16_PEAK_TFLOPS = 2500.0      # MI355X dense f16/bf16 (MI355X_MICROARCH.md, spec)
HBM_PEAK_GBS = 8000.0
HBM_COPY_GBS = 6290.0               # what a float4 copy kernel reaches from HBM (MI355X_MICROARCH.md)


def synthetic_image(seed: int, width: int = 1024, height: int = 1024) -> np.ndarray:
    rng = np.random.default_rng(1000 + seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.zeros((height, width, 4), np.uint8)
    for c in range(3):
        f = np.full((height, width), 128.0, np.float32)
        for _ in range(8):
            fx, fy = rng.uniform(0.002, 0.02, 2)
            f += 14.0 * np.sin(xx * fx + yy * fy + rng.uniform(0, 6.28)).astype(np.float32)
        f += rng.uniform(-8, 8, (height, width)).astype(np.float32)
        img[:, :, c] = np.clip(f, 0, 255).astype(np.uint8)
    img[:, :, 3] = 255
    return img


DECODER_FLOPS = 3.62e9              # per prompt
MIXED_SIZES = [(1800, 1200), (1024, 768), (512, 512), (640, 960), (1024, 1024)]     # config 5


def gemm_shapes(cfg, images: float) -> dict:
    """Algorithmic FLOPs and HBM bytes of ONE launch of each encoder GEMM shape over `images` images stacked in M
    (DESIGN.md section 6): operands once (A [M,K] f16, W [N,K] f16), the residual stream as an f16 pair in and out for the
    stream writers (4 bytes per element each way; the patch embedding reads the fp32 position embedding of ONE image
    instead), the f16 result for the LayerNorm-folded consumers, the per-row statistic partials (8 bytes per row and 256
    columns written by a stream writer, read by a consumer)."""
    M, D, F, PK = images * 4096.0, cfg.embed_dim, cfg.mlp_dim, 768
    pair, stats = 4.0 * M * D, 8.0 * M * (D // 256)
    return {
        "gemm_patch": {"what": "patch embedding", "flops": 2.0 * M * D * PK, "bytes": 2 * M * PK + 2 * D * PK + 4 * 4096 * D + pair + stats},
        "gemm_proj": {"what": "proj (K = D)", "flops": 2.0 * M * D * D, "bytes": 2 * M * D + 2 * D * D + 2 * pair + stats},
        "gemm_fc2": {"what": "fc2 (K = 4 D)", "flops": 2.0 * M * D * F, "bytes": 2 * M * F + 2 * D * F + 2 * pair + stats},
        "gemm_norm": {"what": "qkv", "flops": 2.0 * M * 3 * D * D, "bytes": 2 * M * D + 2 * 3 * D * D + 2 * M * 3 * D + stats},
        "gemm_norm_gelu": {"what": "fc1", "flops": 2.0 * M * F * D, "bytes": 2 * M * D + 2 * F * D + 2 * M * F + stats},
    }


def run_for(fn, threads: int, seconds: float) -> float:
    """Calls fn() -> units from `threads` host threads for `seconds`; returns units per second over all threads."""
    import threading
    fn()
    counts = [0] * threads
    stop = time.perf_counter() + seconds

    def worker(i):
        while time.perf_counter() < stop:
            counts[i] += fn()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    return sum(counts) / (time.perf_counter() - t0)


def config_legs(api, ext, env, cfg, seconds: float) -> dict:
    """BASELINE.json configs 3, 4 and 5 at ONE GPU's share, through the drop-in table with host buffers in and out (PCIe and
    host copies inclusive): slots 13 / 14 with 8 images per call, one caller thread per execution lane.  chip_frac = the
    FLOPs the GPU did per second (encoder + decoder per prompt) / the dense f16 MFMA peak."""
    lanes = ext.lane_count(env)
    enc = cfg.encoder_flops()

    def entry(rate_images, prompts_per_image, what):
        return {"value": rate_images, "unit": "images/s", "masks_per_s": rate_images * prompts_per_image, "caller_threads": lanes,
                "chip_frac": rate_images * (enc + prompts_per_image * DECODER_FLOPS) / 1e12 / MFMA_F16_PEAK_TFLOPS, "workload": what}

    views8 = [api.ImageView(synthetic_image(100 + i), api.Channels.rgba) for i in range(8)]
    pts8 = [api.Point(512, 512)] * 8
    boxes8 = [api.Region(api.Point(256, 256), api.Point(768, 768))] * 8

    def batch8(points=None, regions=None):
        segs = api.Segmentation.process_batch(views8, env)
        api.Segmentation.compute_mask_batch(segs, points=points, regions=regions)
        for sg in segs:
            sg.close()
        return 8

    # config 5 at one GPU's share is ONE batch of 16 mixed-resolution images (sizes cycled) with five prompts each: slot 13
    # takes the 16 images (device resize to 1024, passes of up to four images over the lanes), slot 14 the 80 prompts on the
    # cached embeddings (chunks of eight per launch chain, masks at the images' own resolutions out)
    sizes16 = [MIXED_SIZES[i % len(MIXED_SIZES)] for i in range(16)]
    mixed16 = [api.ImageView(synthetic_image(10 + i, w, h), api.Channels.rgba) for i, (w, h) in enumerate(sizes16)]
    frac5 = ((.5, .5), (.25, .33), (.75, .2), (.6, .8), (.1, .9))
    pts80 = [api.Point(int(w * fx), int(h * fy)) for (w, h) in sizes16 for fx, fy in frac5]

    def mixed16_5prompts():
        segs = api.Segmentation.process_batch(mixed16, env)
        api.Segmentation.compute_mask_batch([sg for sg in segs for _ in range(5)], points=pts80)
        for sg in segs:
            sg.close()
        return 16

    c5_threads = lanes                           # (ViT-B: 641 / 687 / 694 / 705 images/s from 1 / 2 / 3 / 4 caller threads)
    return {
        "config3_share": entry(run_for(lambda: batch8(points=pts8), lanes, seconds), 1,
                               f"configs[2] at one GPU's share: {cfg.name}, 8 images per call (slots 13 / 14), one point each"),
        "config4": entry(run_for(lambda: batch8(regions=boxes8), lanes, seconds), 1,
                         f"configs[3]: {cfg.name}, batch 8 per call (slots 13 / 14), box prompts"),
        "config5_share": {**entry(run_for(mixed16_5prompts, c5_threads, seconds), 5,
                                  f"configs[4] at one GPU's share: {cfg.name}, 16 mixed-resolution images per call {MIXED_SIZES} (cycled) "
                                  "through slot 13 (resized to 1024 on the device), then their 80 prompts -- 5 points per image on "
                                  "the cached embedding -- through slot 14"), "caller_threads": c5_threads},
    }


def device_resident_rate(api, ext, env, steps: int, repeats: int) -> dict:
    """The headline measurement (single-image requests through dlimg_amd_encode_and_mask, everything resident in HBM) for a
    second model in the same run: median over `repeats` blocks of `steps` steps."""
    img = synthetic_image(0)
    p_img, p_mask = ext.device_alloc(env, img.nbytes), ext.device_alloc(env, 1024 * 1024)
    ext.copy_to_device(env, p_img, img)
    views, pts = ext.device_views([p_img], 1024, 1024), [api.Point(512, 512)]
    for _ in range(4):
        ext.encode_and_mask(env, views, pts, [p_mask])
    ext.synchronize(env)
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            ext.encode_and_mask(env, views, pts, [p_mask])
        ext.synchronize(env)
        times.append(time.perf_counter() - t0)
    ext.device_free(env, p_img)
    ext.device_free(env, p_mask)
    dt = float(np.median(times))
    return {"value": steps / dt, "unit": "images/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "repeats": repeats}


def spawn_ranks(n: int, argv: list) -> int:
    """--gpus N from a bare shell: this process has imported neither torch nor the library and has made no GPU call; it starts
    the N ranks as CHILD processes through torch.distributed.run, relays what they print and returns their exit code.
    (Replacing a process that has initialised the GPU by another program takes the machine down on this pool: never exec.)"""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class StubEngine:
    """Host stand-in for the device (--stub-device): a step writes a rank- and step-dependent pattern into this rank's
    mask tensor.  Only the orchestration around it is exercised; no number it produces means anything."""

    def __init__(self, torch, rank: int, batch: int):
        self.torch, self.rank, self.calls = torch, rank, 0
        self.local_masks = torch.zeros((batch, 1024, 1024), dtype=torch.uint8)

    def step(self):
        self.calls += 1
        for i in range(self.local_masks.shape[0]):
            self.local_masks[i, :8, :] = (self.rank * 16 + i + 1) & 0xFF
            self.local_masks[i, 8, 0] = self.calls & 0xFF

    def synchronize(self):
        pass


def inproc_replicas(api, ext, synthetic, model_dir: str, devices: list, images_per_gpu: int, steps: int, rehearsal: bool) -> dict:
    """north_star's own multi-GPU form: the existing C-ABI host, ONE Environment whose replicas sit on `devices`
    (DLIMGEDIT_DEVICES), images dealt one per GPU in turn by slot 13, one prompt each through the device-output form of
    slot 14 (dlimg_amd_get_segmentation_masks_device): every mask is produced on the GPU that holds its embedding and lands
    in ONE buffer on devices[0] (hipMemcpyPeerAsync over xGMI between GPUs).  Host pixels in (PCIe inclusive), masks stay in
    HBM.  Checked against slot 14's host masks, bit for bit, before it is timed."""
    os.environ["DLIMGEDIT_DEVICES"] = ",".join(str(d) for d in devices)
    env = api.Environment(api.Options(api.Backend.gpu, model_dir))
    del os.environ["DLIMGEDIT_DEVICES"]
    try:
        n = images_per_gpu * len(devices)
        imgs = [synthetic(500 + i) for i in range(n)]
        views = [api.ImageView(im, api.Channels.rgba) for im in imgs]
        pts = [api.Point(512, 512)] * n
        out = ext.device_alloc(env, n * 1024 * 1024)
        segs = api.Segmentation.process_batch(views, env)
        placement = [ext.segmentation_device(sg)[0] for sg in segs]
        want = api.Segmentation.compute_mask_batch(segs, points=pts)
        offsets = ext.compute_mask_batch_device(segs, out, points=pts, root_device=devices[0])
        got = np.empty(n * 1024 * 1024, np.uint8)
        ext.copy_to_host(env, got, out)
        equal = all(np.array_equal(got[offsets[i]:offsets[i] + 1024 * 1024].reshape(1024, 1024), want[i]) for i in range(n))
        for sg in segs:
            sg.close()

        def one_call():
            sg_ = api.Segmentation.process_batch(views, env)
            ext.compute_mask_batch_device(sg_, out, points=pts, root_device=devices[0])
            for x in sg_:
                x.close()

        one_call()
        t0 = time.perf_counter()
        for _ in range(steps):
            one_call()
        dt = time.perf_counter() - t0
        ext.device_free(env, out)
        return {"value": n * steps / dt, "unit": "images/s", "devices": devices, "replicas": ext.replica_count(env),
                "images_per_call": n, "calls": steps, "images_per_replica": [placement.count(r) for r in range(len(devices))],
                "gathered_masks_equal_slot_14": bool(equal),
                "note": ("one-GPU REHEARSAL (the same GPU listed %d times), not a result" % len(devices)) if rehearsal else
                        "one process, one Environment over all GPUs: host pixels in through slot 13, masks gathered into one "
                        "buffer on devices[0] by dlimg_amd_get_segmentation_masks_device (peer copies over xGMI); one host "
                        "thread, PCIe inclusive"}
    finally:
        env.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=21, help="timed repeats of the block of --steps steps (median reported)")
    ap.add_argument("--no-abi-path", action="store_true")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="vit_b")
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the BASELINE configs 3-5 / ViT-H legs (`configs`)")
    ap.add_argument("--leg-seconds", type=float, default=2.0, help="duration of each leg of `configs`")
    ap.add_argument("--model-dir", default=None, help="existing model directory (default: seeded synthetic weights)")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="one-GPU rehearsal of the N > 1 code path: every rank uses GPU 0 and the collectives run on gloo / CPU "
                         "tensors (RCCL refuses two ranks on one device); the line is marked as a rehearsal, never a result")
    ap.add_argument("--stub-device", action="store_true",
                    help="no GPU at all: a host stand-in takes the device's place so that the N > 1 orchestration runs on "
                         "CPU (gloo); the line is marked, its numbers mean nothing (tests/test_bench_cpu.py)")
    ap.add_argument("--inproc-images-per-gpu", type=int, default=8, help="images per GPU per call of the inproc_replicas leg")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))      # before anything here has touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    stub = args.stub_device
    host_collectives = args.rehearse_gloo or stub
    dev_index = 0 if host_collectives else local_rank
    coll_dev = "cpu" if host_collectives else "cuda"
    os.environ["DLIMGEDIT_DEVICE"] = str(dev_index)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
        torch.cuda.set_device(dev_index)
    if world > 1:
        if host_collectives:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # a host-side group for waiting: a rank parked in an RCCL barrier keeps a polling kernel on its GPU
    host_group = dist.new_group(backend="gloo") if world > 1 and not host_collectives else None

    def device_synchronize():
        if not stub:
            torch.cuda.synchronize()

    from dlimgedit_amd import sharding
    from dlimgedit_amd.sam_config import get_config
    if not stub:
        from dlimgedit_amd import api, weights as W

    cfg = get_config(args.model)
    # ---- model directory: rank 0 of the node writes seeded synthetic weights, the others wait
    model_dir = args.model_dir
    params = None
    if model_dir is None and not stub:
        model_dir = os.path.join(tempfile.gettempdir(), f"dlimgedit_bench_{args.model}_{args.seed}_{os.getuid()}")
        target = Path(model_dir) / "segmentation" / W.weight_file_name(cfg)
        if local_rank == 0 and not target.exists():
            params = W.synthetic_weights(cfg, args.seed)
            tmp = target.with_suffix(".tmp")
            W.save_weights(tmp, cfg, params)
            os.replace(tmp, target)
    if world > 1:
        dist.barrier()
    os.environ["DLIMGEDIT_SAM_MODEL"] = args.model

    B = args.batch
    img_ptrs, mask_ptrs = [], []
    if stub:
        env = ext = None
        engine = StubEngine(torch, rank, B)
        local_masks = engine.local_masks
        step, engine_sync = engine.step, engine.synchronize
    else:
        env = api.Environment(api.Options(api.Backend.gpu, model_dir))
        ext = api.ext
        # ---- inputs resident in HBM before the timed region
        imgs = [synthetic_image(rank * B + i) for i in range(B)]
        # N > 1: the masks live in ONE torch tensor per rank, so the RCCL gather below reads them where the kernels wrote
        # them (no host round trip); N = 1 keeps torch off the data path altogether
        local_masks = torch.zeros((B, 1024, 1024), dtype=torch.uint8, device="cuda") if world > 1 else None
        for i, im in enumerate(imgs):
            p = ext.device_alloc(env, im.nbytes)
            ext.copy_to_device(env, p, im)
            img_ptrs.append(p)
            mask_ptrs.append(local_masks[i].data_ptr() if world > 1 else ext.device_alloc(env, 1024 * 1024))
        views = ext.device_views(img_ptrs, 1024, 1024)
        points = [api.Point(512, 512)] * B

        def step():
            ext.encode_and_mask(env, views, points, mask_ptrs)

        def engine_sync():
            ext.synchronize(env)

    def sync_all():
        engine_sync()
        device_synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    repeat_s, own_s = [], []
    for _ in range(max(1, args.repeats)):
        sync_all()                               # barrier + synchronize in front of the timed steps ...
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        engine_sync()
        device_synchronize()                     # ... and behind them
        own = time.perf_counter() - t0
        own_s.append(own)
        repeat_s.append(sharding.max_over_ranks(own, device=coll_dev))
        if world > 1:
            dist.barrier()
    elapsed = float(np.median(repeat_s))
    # every rank's own clock of the median repeat (N > 1): the MAX above is what `value` uses, this shows who set it
    median_repeat = int(np.argsort(repeat_s)[len(repeat_s) // 2])
    per_rank_s = sharding.all_ranks(own_s[median_repeat], device=coll_dev) if world > 1 else [own_s[median_repeat]]

    decoder_flops = DECODER_FLOPS                # per prompt
    result = None
    if rank == 0:
        images = world * B * args.steps
        result = {
            "metric": "images/sec encode+mask @1024x1024",
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"SAM {args.model} encoder + 1 point prompt, {B} image(s)/GPU/step, 1024x1024 RGBA, "
                                   "inputs and masks resident in HBM; timed through the extension entry point "
                                   "dlimg_amd_encode_and_mask (device pointers in and out) -- the drop-in table itself, host "
                                   "buffers in and out, is `abi_path`", "images_per_gpu_per_step": B,
                       "weights": "seeded synthetic" if args.model_dir is None else "from --model-dir"},
            "repeats": len(repeat_s),
            "timed_total_s": float(sum(repeat_s)),
            "repeat_ms": [round(1e3 * t, 3) for t in repeat_s],
            "value_min_max": [images / max(repeat_s), images / min(repeat_s)],
            "per_rank_value": [B * args.steps / t for t in per_rank_s],
        }
        if stub:
            result["stub_device"] = "no GPU: a host stand-in took the device's place (orchestration rehearsal); the numbers mean nothing"
            result["data"] = "none (stub device)"
        else:
            # what the library's step queue really uses (it clamps the variables and has its own defaults)
            result["config"].update({"lanes_per_gpu": ext.lane_count(env), "lanes_fed_by_the_step_queue": ext.queue_config(env)["lanes_in_use"],
                                     "requests_coalesced_per_pass": ext.queue_config(env)["coalesce"],
                                     "passes_queued_per_lane": ext.queue_config(env)["step_depth"]})

    # ---- N > 1: the ranks that really take part, and the optional gather of the masks (after the timed region)
    if world > 1:
        ranks = sharding.count_ranks(device=coll_dev)
        engine_sync()                            # the masks of the last timed step are in local_masks (device memory)
        device_synchronize()
        foreground = float((local_masks > 0).float().mean().item())
        src = local_masks.cpu() if host_collectives else local_masks
        sharding.gather_device_masks(src, world * B)                     # first call: communicator set-up, not timed
        device_synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        gathered = sharding.gather_device_masks(src, world * B)
        device_synchronize()
        t_gather = sharding.max_over_ranks(time.perf_counter() - t0, device=coll_dev)
        ok = bool(torch.equal(gathered[rank::world][:B], src))             # item i = b * world + rank
        all_ok = sharding.max_over_ranks(0.0 if ok else 1.0, device=coll_dev) == 0.0
        if ranks != world or not all_ok:
            # a multi-GPU line is only printed when every rank took part in the collective and got its own masks back
            if rank == 0:
                print(json.dumps({"error": "multi-GPU check failed", "rccl_ranks": ranks, "expected_ranks": world,
                                  "own_share_intact_on_every_rank": all_ok}), file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)
        if rank == 0:
            result["rccl"] = {"rccl_ranks": ranks, "backend": "gloo (no device: orchestration REHEARSAL on CPU, not a result)" if stub else
                              "gloo (one-GPU REHEARSAL, not a result)" if args.rehearse_gloo else "nccl (RCCL)",
                              "gather": {"masks": int(gathered.shape[0]), "bytes": int(gathered.numel()), "ms": 1e3 * t_gather,
                                         "gbs": gathered.numel() / t_gather / 1e9, "own_share_intact": ok,
                                         "own_share_foreground": foreground,
                                         "note": "RCCL all_gather_into_tensor straight from the device buffers the post-processing "
                                                 "kernel wrote (no host copy), outside the timed region: no collective is on "
                                                 "the data path"}}
    elif rank == 0:
        result["rccl"] = {"rccl_ranks": 1, "gather": None}

    # ---- N > 1: the node driven from ONE process through the drop-in table (rank 0; the other ranks wait on the HOST)
    if world > 1 and not stub:
        if rank == 0:
            devices = [0] * world if args.rehearse_gloo else list(range(world))
            try:
                result["inproc_replicas"] = inproc_replicas(api, ext, synthetic_image, model_dir, devices, args.inproc_images_per_gpu,
                                                            max(2, min(args.steps, 10)), args.rehearse_gloo)
            except Exception as e:                       # the torchrun figure above stands on its own
                result["inproc_replicas"] = {"error": str(e)}
        dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    # ---- profiled repeats of the same steps: HIP events attached to every GEMM dispatch, on the stream it is launched on.
    # First in the regime `value` is measured in (all lanes, requests coalesced as in the timed region), then with every
    # request on lane 0 (each kernel alone on the chip).
    if rank == 0 and not stub:
        def profiled(mode):
            ext.set_profiling(env, mode)
            ext.take_stage_stats(env)
            for _ in range(args.steps):
                step()
            ext.synchronize(env)
            stats = ext.take_stage_stats(env)
            ext.set_profiling(env, 0)
            return stats

        st_lanes = profiled(2)
        st = profiled(1)
        g, gl = st["gemm"], st_lanes["gemm"]
        # Every GEMM SHAPE of the encoder has its own clock (r06: patch / proj / fc2 share a kernel and a grid but not an
        # arithmetic intensity -- proj at ViT-B moves a byte per 152 FLOPs, below the ridge of 2500 TFLOP/s : 8 TB/s = 312).
        flavours = ("gemm_stats", "gemm_norm", "gemm_norm_gelu", "gemm_other", "gemm_patch", "gemm_proj", "gemm_fc2")   # sub-clocks of "gemm"
        kernels = {"gemm_patch": "gemm_pp_kernel<ACT_NONE, EPI_STATS>: patch embedding (bias + position embedding in, residual "
                                 "stream as an f16 pair + row statistics out)",
                   "gemm_proj": "gemm_pp_kernel<ACT_NONE, EPI_STATS>: proj (bias + stream pair in, pair + row statistics out)",
                   "gemm_fc2": "gemm_pp_kernel<ACT_NONE, EPI_STATS>: fc2 (the same, K = 4 D)",
                   "gemm_norm": "gemm_pp_kernel<ACT_NONE, EPI_NORM>: qkv (LayerNorm folded in, f16 out)",
                   "gemm_norm_gelu": "gemm_pp_kernel<ACT_GELU, EPI_NORM>: fc1 (LayerNorm folded in, GELU, f16 out)",
                   "gemm_other": "gemm_f16_kernel: neck 1x1 / 3x3"}
        ridge = MFMA_F16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)          # FLOP per byte where the two roofs meet

        def rate(s_):       # TFLOP/s of a stage record
            return s_["work"] / (s_["ms"] * 1e-3) / 1e12 if s_["ms"] > 0 else 0.0

        per_kernel = {}
        for key, what in kernels.items():
            a1, al = st[key], st_lanes[key]
            if a1["launches"] == 0:
                continue
            e = {"kernel": what, "launches_per_step": a1["launches"] / args.steps,
                 "gflop_per_launch": a1["work"] / a1["launches"] / 1e9,
                 "avg_launch_us": 1e3 * a1["ms"] / a1["launches"], "tflops": rate(a1),
                 "frac_mfma": rate(a1) / MFMA_F16_PEAK_TFLOPS,
                 "share_of_gemm_time": a1["ms"] / g["ms"] if g["ms"] > 0 else 0.0,
                 "under_lanes_avg_launch_us": 1e3 * al["ms"] / max(1, al["launches"])}
            shape1 = gemm_shapes(cfg, 1.0).get(key)
            if shape1:
                # images stacked in an average launch of this shape (whole passes: 4 at ViT-B, 2 at the larger models)
                images = (a1["work"] / a1["launches"]) / shape1["flops"]
                sh = gemm_shapes(cfg, images)[key]
                gbs = sh["bytes"] / (e["avg_launch_us"] * 1e-6) / 1e9
                e.update({"images_per_launch": images, "algorithmic_mb_per_launch": sh["bytes"] / 1e6, "gbs": gbs,
                          "frac_hbm": gbs / HBM_PEAK_GBS, "flop_per_byte": sh["flops"] / sh["bytes"],
                          "bound": "hbm" if sh["flops"] / sh["bytes"] < ridge else "mfma"})
                e["frac"] = e["frac_hbm"] if e["bound"] == "hbm" else e["frac_mfma"]
            else:
                e.update({"bound": "mfma", "frac": e["frac_mfma"]})
            per_kernel[key] = e
        dominant = max(per_kernel, key=lambda k_: per_kernel[k_]["share_of_gemm_time"]) if per_kernel else None
        dom = per_kernel.get(dominant, {"tflops": 0.0, "frac": 0.0, "frac_mfma": 0.0, "avg_launch_us": 0.0, "gflop_per_launch": 0.0,
                                        "kernel": "-", "bound": "mfma"})
        achieved_alone = rate(g)
        # HBM-side bytes per launch of the dominant kernel are NOT measured in this run: counters cannot be read inside it.
        # They come from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE) -- and only from a profile whose recorded command names this
        # model and whose table has the dominant kernel AT THE GRID it ran with here; otherwise `traffic` stays null.
        traffic, traffic_source = None, None

        def round_of(path):
            import re
            m = re.match(r"r(\d+)_", path.name)
            return int(m.group(1)) if m else -1
        newest = sorted((ROOT / "profiles").glob("r*_hbm_traffic_pmc.json"), key=round_of)
        if newest and B == 1 and dominant and "images_per_launch" in dom:
            doc = json.loads(newest[-1].read_text())
            tag, ncols = {"gemm_patch": ("gemm_pp_kernel<0,2>", cfg.embed_dim), "gemm_proj": ("gemm_pp_kernel<0,2>", cfg.embed_dim),
                          "gemm_fc2": ("gemm_pp_kernel<0,2>", cfg.embed_dim), "gemm_norm": ("gemm_pp_kernel<0,1>", 3 * cfg.embed_dim),
                          "gemm_norm_gelu": ("gemm_pp_kernel<1,1>", cfg.mlp_dim)}.get(dominant, (None, 0))
            wgs = int(round(dom["images_per_launch"])) * 16 * (ncols // 256)
            by_shape = doc.get("by_shape", {}).get(dominant)              # r06 profiles: the stream writers told apart
            row_key = f"{tag} wgs={wgs}"
            row = by_shape if by_shape and by_shape.get("grid") == wgs else doc.get("by_kernel_and_grid", {}).get(row_key)
            shared_row = dominant in ("gemm_patch", "gemm_proj", "gemm_fc2") and row is not by_shape
            import re
            doc_model = re.search(r"--model (\w+)", doc.get("command", ""))
            if row and (doc_model.group(1) if doc_model else "vit_b") == args.model and not shared_row:
                traffic = row["fetch_bytes_per_launch_corrected_x2"] + row["write_bytes_per_launch"]
                traffic_source = (f"profiles/{newest[-1].name}, row '{dominant if row is by_shape else row_key}': FETCH_SIZE (doubled, "
                                  "gfx950) + WRITE_SIZE per launch from separate rocprofv3 --pmc passes of this command, committed "
                                  "with the round -- not measured inside this run")
        step_flops = B * (cfg.encoder_flops() + decoder_flops)
        chip_tflops = step_flops / (result["ms_per_step"] * 1e-3) / 1e12          # per GPU (every rank runs the same step)
        lanes_wall_ms = result["ms_per_step"] * args.steps
        hbm_bound = dom.get("bound") == "hbm"
        result["roofline"] = {
            # the dominant kernel = the GEMM shape with the largest share of GPU time, clocked ALONE on the chip (single
            # lane): algorithmic FLOPs (or bytes, when its arithmetic intensity puts it under the HBM roof) of its launches /
            # their own dispatch-to-completion time.  A per-kernel rocprofv3 table reproduces it
            # (profiles/*_kernel_stats_single_lane_by_grid.txt; the stream writers' shapes: profiles/*_gemm_shapes.txt).
            "kernel": dom["kernel"], "kernel_key": dominant,
            "bound": dom.get("bound", "mfma"),
            "achieved": dom.get("gbs", 0.0) if hbm_bound else dom["tflops"],
            "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F16_PEAK_TFLOPS, "unit": "GB/s" if hbm_bound else "TFLOP/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_source,
            "traffic_note": "FETCH_SIZE counts the L2's requests to the fabric, Infinity-Cache hits included (MI355X_MICROARCH.md): for a "
                            "LayerNorm-folded consumer (qkv, fc1) it is 4-5 x the algorithmic reads because every round of 32 concurrent "
                            "tiles of an XCD fetches its row and column panels again -- 100 MB of output stream through the 4 MB L2 "
                            "between rounds -- while the operands themselves (25 MB of activations, < 5 MB of weights) stay in the "
                            "256 MB Infinity Cache; the model (panels per round x 393 KB x 8 XCDs) reproduces the counters within 4 % "
                            "(LABNOTES r06).  WRITE_SIZE equals the algorithmic output.",
            "bound_rule": f"arithmetic intensity of the launch (algorithmic FLOPs / algorithmic bytes, DESIGN.md section 6) against the "
                          f"ridge {ridge:.0f} FLOP/B = {MFMA_F16_PEAK_TFLOPS:.0f} TFLOP/s : {HBM_PEAK_GBS / 1e3:.0f} TB/s; every shape "
                          "carries both fractions in per_kernel",
            "mode": "single lane: every request on lane 0, each kernel alone on the chip; passes of as many images as in the timed region",
            "avg_launch_us": dom["avg_launch_us"], "gflop_per_launch": dom["gflop_per_launch"],
            "algorithmic_mb_per_launch": dom.get("algorithmic_mb_per_launch"),
            "per_kernel": per_kernel,
            "all_gemm_launches": {"achieved_single_lane": achieved_alone, "frac_single_lane": achieved_alone / MFMA_F16_PEAK_TFLOPS,
                                  "avg_launch_us_single_lane": 1e3 * g["ms"] / max(1, g["launches"]), "launches": g["launches"]},
            "achieved_single_lane": achieved_alone, "frac_single_lane": achieved_alone / MFMA_F16_PEAK_TFLOPS,
            "clock": "HIP events attached to each GEMM dispatch (hipExtLaunchKernelGGL) on the lane's own stream: the kernel's "
                     "own execution time",
            # whole chip, regime of `value`: the figure the driver's clock supports
            "chip_achieved": chip_tflops, "chip_frac": chip_tflops / MFMA_F16_PEAK_TFLOPS,
            "chip_note": "per GPU: (encoder + decoder FLOPs of a step) / ms_per_step / peak -- all lanes, every kernel, gaps included",
            # all lanes running: a launch is clocked while it SHARES the chip, so this is not a roofline fraction of anything;
            # kept as a contention diagnostic only
            "under_lanes": {"avg_gemm_launch_us": 1e3 * gl["ms"] / max(1, gl["launches"]),
                            "gemm_kernels_in_flight": gl["ms"] / lanes_wall_ms if lanes_wall_ms > 0 else 0.0,
                            # every stage's own clocks with all lanes running (HIP events on the lanes' streams, no profiler:
                            # rocprofv3's kernel trace runs the lanes' passes one after another on this pool, r04)
                            "per_stage": {name: {"launches_per_step": v["launches"] / args.steps,
                                                 "avg_launch_us": 1e3 * v["ms"] / v["launches"],
                                                 "ms_per_step": v["ms"] / args.steps,
                                                 "slowdown_vs_single_lane": (v["ms"] / v["launches"]) / (st[name]["ms"] / st[name]["launches"])
                                                 if st[name]["launches"] and st[name]["ms"] > 0 else None}
                                          for name, v in st_lanes.items() if v["launches"] and name not in flavours},
                            "kernels_in_flight": sum(v["ms"] for name, v in st_lanes.items() if name not in flavours) / lanes_wall_ms
                            if lanes_wall_ms > 0 else 0.0,
                            "note": "sum of the GEMM launches' durations / wall time of the profiled steps = GEMM kernels in "
                                    "flight on average; each holds a share of the CUs while it is clocked"},
        }
        stages = {}
        for name, s in st.items():
            if s["launches"] == 0 or name in flavours:
                continue
            per_step_ms = s["ms"] / args.steps
            e = {"ms_per_step": per_step_ms, "launches_per_step": s["launches"] / args.steps}
            if name in ("gemm", "attention_window", "attention_global", "decoder"):
                e["tflops"] = s["work"] / (s["ms"] * 1e-3) / 1e12
            else:
                e["gbs"] = s["work"] / (s["ms"] * 1e-3) / 1e9
                e["frac_hbm_peak"] = e["gbs"] / HBM_PEAK_GBS
            stages[name] = e
        result["stages"] = stages
        result["profile_mode"] = ("`stages`: single lane, serial -- a repeat of the timed steps with every request on lane 0, so "
                                  "each kernel runs alone on the chip; GEMM launches are clocked by events attached to the "
                                  "dispatch, the other stages by hipEventRecord pairs around the launch (which adds a few "
                                  "microseconds per launch).  The stage times therefore do not add up to ms_per_step, "
                                  "which overlaps the lanes.  `roofline` is the dominant GEMM flavour of this same single-lane repeat; "
                                  "`roofline.under_lanes` comes from a second repeat with all lanes.")
        enc_ms = sum(v["ms_per_step"] for k, v in stages.items() if k not in ("decoder", "post", "pre"))
        result["encoder"] = {"gflop_per_image": cfg.encoder_flops() / 1e9, "event_ms_per_step": enc_ms,
                             "tflops": B * cfg.encoder_flops() / (enc_ms * 1e-3) / 1e12 if enc_ms > 0 else 0.0,
                             "mfma_frac": B * cfg.encoder_flops() / (enc_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS
                             if enc_ms > 0 else 0.0}

    # ---- per-stage rates SURVEY.md section 8d asks for next to the metric: encoder alone (device resident, all lanes)
    if rank == 0 and world == 1 and not stub:
        for _ in range(2):
            ext.encode_only(env, views)
        ext.synchronize(env)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ext.encode_only(env, views)
        ext.synchronize(env)
        result["encoder_only"] = {"value": B * args.steps / (time.perf_counter() - t0), "unit": "images/s",
                                  "note": "pre-processing + ViT encoder + neck, inputs resident in HBM, all lanes"}

    # ---- the two pixel kernels against the HBM rate (north_star: "HBM GB/s for the pre/post kernels"): one image per
    # launch sits on the launch floor (`stages` above), so they are also clocked with 16 images / masks per launch.
    # Launches rotate over 768 MB of distinct inputs and outputs (3 x the 256 MB Infinity Cache): what moves is HBM traffic.
    if rank == 0 and world == 1 and not stub:
        pre_b, post_b = 4 * 1024 * 1024 + 4096 * 768 * 2, 256 * 256 * 4 + 1024 * 1024      # algorithmic bytes per image / mask
        hk = {"unit": "GB/s", "peak": HBM_PEAK_GBS, "achievable": HBM_COPY_GBS,
              "achievable_note": "float4 copy kernel, HBM to HBM (MI355X_MICROARCH.md)",
              "working_set_mb": 768, "clock": "HIP events around 60 back-to-back launches on one stream, each on the next "
              "set of a ring of distinct inputs and outputs (768 MB footprint, nothing Infinity-Cache resident)",
              "bytes_per_image": {"pre": pre_b, "post": post_b},
              "library": "lib/libdlimgedit_test.so (dlimg_amd_bench_prepost: the product's kernel objects behind a timing loop "
                         "that the product library does not export)"}
        for n in (1, 16):
            pre_ms, post_ms = ext.bench_prepost(n, 60, 768)
            e = {"pre_us": 1e3 * pre_ms, "pre_gbs": n * pre_b / (pre_ms * 1e-3) / 1e9,
                 "post_us": 1e3 * post_ms, "post_gbs": n * post_b / (post_ms * 1e-3) / 1e9}
            for kname in ("pre", "post"):
                e[f"{kname}_frac"] = e[f"{kname}_gbs"] / HBM_PEAK_GBS
                e[f"{kname}_frac_of_achievable"] = e[f"{kname}_gbs"] / HBM_COPY_GBS
            hk[f"batch{n}"] = e
        result["hbm_kernels"] = hk

    # ---- the drop-in ABI itself: host buffers in and out (PCIe inclusive), rank 0, N = 1 only
    if rank == 0 and world == 1 and not stub and not args.no_abi_path:
        import threading
        view = api.ImageView(imgs[0], api.Channels.rgba)

        def one_image():
            seg = api.Segmentation.process(view, env)
            seg.compute_mask(api.Point(512, 512))
            seg.close()
            return 1

        # the same with the pixels held as the reference's wrapper holds a loaded image: in an Image, i.e. in memory from the
        # table's create_image / load_image, which this library pins and reads in place (csrc/image_memory.hpp); the result
        # masks are Images in both forms (api.py mirrors the wrapper's compute_mask), written in place
        held = api.Image(api.Extent(imgs[0].shape[1], imgs[0].shape[0]), api.Channels.rgba)
        held.pixels()[...] = imgs[0]
        held_view = held.view()

        def one_image_held():
            seg = api.Segmentation.process(held_view, env)
            seg.compute_mask(api.Point(512, 512))
            seg.close()
            return 1

        # slot 3 alone, one thread: process() returns once its pass is enqueued (csrc/segmentation.hpp), so a loop over images
        # keeps the lanes fed by itself; a handle is closed (which waits for its pass) eight images later
        open_handles = []

        def process_only():
            open_handles.append(api.Segmentation.process(view, env))
            if len(open_handles) > 8:
                open_handles.pop(0).close()
            return 1

        views8 = [api.ImageView(synthetic_image(100 + i), api.Channels.rgba) for i in range(8)]
        pts8 = [api.Point(512, 512)] * 8

        def batch8():
            segs = api.Segmentation.process_batch(views8, env)
            api.Segmentation.compute_mask_batch(segs, points=pts8)
            for sg in segs:
                sg.close()
            return 8

        def rate(fn, threads, seconds=1.5):
            fn()
            counts = [0] * threads
            stop = time.perf_counter() + seconds

            def worker(i):
                while time.perf_counter() < stop:
                    counts[i] += fn()
            ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
            t0 = time.perf_counter()
            [t.start() for t in ts]
            [t.join() for t in ts]
            return sum(counts) / (time.perf_counter() - t0)

        cached = api.Segmentation.process(view, env)

        def one_prompt():
            cached.compute_mask(api.Point(512, 512))
            return 1

        def five_prompts():
            api.Segmentation.compute_mask_batch([cached] * 5, points=[api.Point(200 + 150 * j, 300 + 100 * j) for j in range(5)])
            return 5

        lanes = ext.lane_count(env)
        result["decode_only"] = {
            "unit": "prompts/s", "note": "prompt encoder + mask decoder + post-processing on a cached embedding through "
                                         "the ABI (slots 4 / 14), 1 MiB host mask out per prompt",
            "one_prompt_per_call_one_thread": rate(one_prompt, 1, 1.0),
            f"one_prompt_per_call_{lanes}_threads": rate(one_prompt, lanes, 1.0),
            "five_prompts_per_call_one_thread": rate(five_prompts, 1, 1.0),
            f"five_prompts_per_call_{lanes}_threads": rate(five_prompts, lanes, 1.0),
        }
        cached.close()
        result["abi_path"] = {
            "unit": "images/s", "note": "host pixels in, host masks out through dlimg_Api; PCIe and host copies included; pixels in a numpy buffer "
                                        "of the program's own (staged through the library's pinned ring) unless the key says "
                                        "otherwise, result masks in Images from create_image as the reference's wrapper makes them",
            "slots_3_4_one_thread": rate(one_image, 1),
            "slots_3_4_one_thread_pixels_in_an_Image": rate(one_image_held, 1),
            f"slots_3_4_{lanes}_threads": rate(one_image, lanes),
            "slots_13_14_batch8_one_thread": rate(batch8, 1),
            "slots_13_14_batch8_two_threads": rate(batch8, 2),
            "slot_3_only_one_thread": rate(process_only, 1),
        }
        for sg in open_handles:
            sg.close()

    # ---- BASELINE.json configs 3-5 at one GPU's share and the ViT-H model, in the driver's own run: the
    # headline above is configs[1]; these legs are short (--leg-seconds each) and run after it, rank 0, N = 1 only
    if rank == 0 and world == 1 and not stub and not args.no_config_legs:
        legs = {args.model: config_legs(api, ext, env, cfg, args.leg_seconds)}
        if args.model == "vit_b" and args.model_dir is None:
            # configs[3] / configs[4] name the ViT-H encoder: a second Environment on seeded synthetic ViT-H weights (this
            # process writes them once: ~40 s of host time on a fresh box), its device-resident rate measured like `value`
            # and the same three legs through the drop-in table
            h_cfg = get_config("vit_h")
            h_dir = os.path.join(tempfile.gettempdir(), f"dlimgedit_bench_vit_h_{args.seed}_{os.getuid()}")
            h_target = Path(h_dir) / "segmentation" / W.weight_file_name(h_cfg)
            t_w = time.perf_counter()
            if not h_target.exists():
                h_tmp = h_target.with_suffix(".tmp")
                W.save_weights(h_tmp, h_cfg, W.synthetic_weights(h_cfg, args.seed))
                os.replace(h_tmp, h_target)
            t_w = time.perf_counter() - t_w
            os.environ["DLIMGEDIT_SAM_MODEL"] = "vit_h"
            h_env = api.Environment(api.Options(api.Backend.gpu, h_dir))
            try:
                resident = device_resident_rate(api, ext, h_env, args.steps, 5)
                resident["chip_frac"] = resident["value"] * (h_cfg.encoder_flops() + DECODER_FLOPS) / 1e12 / MFMA_F16_PEAK_TFLOPS
                resident["workload"] = ("SAM vit_h encoder + 1 point prompt, 1 image per step, inputs and masks resident in HBM, through "
                                        "dlimg_amd_encode_and_mask: the headline's measurement on the ViT-H model")
                resident["queue"] = ext.queue_config(h_env)
                legs["vit_h"] = {"device_resident": resident, **config_legs(api, ext, h_env, h_cfg, args.leg_seconds),
                                 "weights_written_s": t_w}
            finally:
                h_env.close()
                os.environ["DLIMGEDIT_SAM_MODEL"] = args.model
        result["configs"] = {"note": "BASELINE.json configs[2..4] at ONE GPU's share through the drop-in table (host pixels in, host masks "
                                     "out: PCIe and host copies inclusive), one caller thread per execution lane; `value` above is "
                                     "configs[1].  chip_frac = FLOPs done per second / 2500 TFLOP/s", **legs}

    # ---- CPU baseline (oracle port) + mask IoU, rank 0, N = 1 only
    if rank == 0 and world == 1 and not stub and not args.no_cpu_baseline:
        from oracle import sam_oracle as O
        if params is None:
            params = W.load_weights(Path(model_dir) / "segmentation" / W.weight_file_name(cfg))[1] \
                if args.model_dir else W.synthetic_weights(cfg, args.seed)
        gpu_mask = np.empty((1024, 1024), np.uint8)
        ext.copy_to_host(env, gpu_mask, mask_ptrs[0])
        # BLAS threads: a one-GPU box has a CPU quota of 16 (cgroup cpu.max) whatever os.cpu_count() says (256); more
        # threads than that oversubscribe (per image: 15 s with 256 threads, 10.7 s with 32, 7.5 s with 16, 7.6 s with 8)
        threads = min(16, os.cpu_count() or 1)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            ora = O.OracleSegmentation(params, cfg).process(imgs[0], O.CH_RGBA)
            t_enc = time.perf_counter() - t0
            cpu_mask = ora.compute_mask(point=(512, 512))
            t_all = time.perf_counter() - t0

        def iou_of(a, b):
            union = np.logical_or(a > 0, b > 0).sum()
            return float(np.logical_and(a > 0, b > 0).sum()) / float(union) if union else 1.0

        result["cpu_baseline"] = {"value": 1.0 / t_all, "unit": "images/s", "cores": threads, "kind": "port",
                                  "sample": f"1 image of the same workload ({args.model} encode {t_enc:.1f} s + 1 point "
                                            f"mask {t_all - t_enc:.2f} s), numpy fp32 oracle, {threads} BLAS threads "
                                            f"of {os.cpu_count()} host cores"}
        # mask IoU vs the CPU oracle: the timed prompt plus further prompts through the drop-in ABI
        seg = api.Segmentation.process(api.ImageView(imgs[0], api.Channels.rgba), env)
        checks = [{"prompt": "point(512,512) [timed step]", "iou": iou_of(gpu_mask, cpu_mask),
                   "foreground": float((cpu_mask > 0).mean())}]
        for name, gp, op in (("point(200,800)", api.Point(200, 800), dict(point=(200, 800))),
                             ("point(800,200)", api.Point(800, 200), dict(point=(800, 200))),
                             ("box(256,256,768,768)", api.Region(api.Point(256, 256), api.Point(768, 768)),
                              dict(region=(256, 256, 768, 768)))):
            want = ora.compute_mask(**op)
            checks.append({"prompt": name, "iou": iou_of(seg.compute_mask(gp), want), "foreground": float((want > 0).mean())})
        low_gpu, iou_gpu = ext.get_logits(seg, point=api.Point(512, 512))
        low_cpu, iou_cpu = ora.logits(point=(512, 512))
        result["logits_check"] = {"max_abs_error": float(np.abs(low_gpu - low_cpu).max()),
                                  "max_abs_logit": float(np.abs(low_cpu).max()),
                                  "fraction_abs_logit_below_1e-3": float((np.abs(low_cpu) < 1e-3).mean()),
                                  "iou_prediction_max_abs_error": float(np.abs(iou_gpu - iou_cpu).max()),
                                  "embedding_max_abs_error": float(np.abs(ext.get_embedding(seg) - ora.embedding).max())}
        result["mask_iou"] = min(c["iou"] for c in checks)
        result["mask_iou_checks"] = checks

    if rank == 0:
        print(json.dumps(result), flush=True)
    if not stub:
        for p in img_ptrs + (mask_ptrs if world == 1 else []):
            ext.device_free(env, p)
        env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
