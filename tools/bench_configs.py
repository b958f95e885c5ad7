"""Throughput of BASELINE.json configs 2-5 THROUGH THE DROP-IN ABI (host pixel buffers in, host masks out:
PCIe-inclusive, synchronous calls, one host thread per execution lane).  One GPU; the multi-GPU configs are
their per-GPU share.  python tools/bench_configs.py [vit_b|vit_h] [seconds]"""
import sys, tempfile, threading, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config

variant = sys.argv[1] if len(sys.argv) > 1 else "vit_b"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
cfg = get_config(variant)


def run_threads(n_threads, fn):
    """fn(thread_index) -> images processed; returns images/s over all threads."""
    counts = [0] * n_threads
    stop = time.perf_counter() + budget
    def worker(i):
        while time.perf_counter() < stop:
            counts[i] += fn(i)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]
    return sum(counts) / (time.perf_counter() - t0)


with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    lanes = api.ext.lane_count(env)
    img = synthetic_image(0)
    view = api.ImageView(img, api.Channels.rgba)
    api.Segmentation.process(view, env).compute_mask(api.Point(512, 512))      # warm-up / model load

    def one_image(_):
        api.Segmentation.process(view, env).compute_mask(api.Point(512, 512))
        return 1
    print(f"{variant}: config 2 (batch 1, 1 point), ABI path, 1 thread : {run_threads(1, one_image):8.1f} img/s")
    print(f"{variant}: config 2, ABI path, {lanes} threads (one per lane)   : {run_threads(lanes, one_image):8.1f} img/s")

    views8 = [api.ImageView(synthetic_image(i), api.Channels.rgba) for i in range(8)]
    pts8 = [api.Point(512, 512)] * 8
    def batch8_points(_):
        segs = api.Segmentation.process_batch(views8, env)
        api.Segmentation.compute_mask_batch(segs, points=pts8)
        return 8
    print(f"{variant}: config 3 share (batch 8/GPU, 1 point each), ABI batch entry points, {lanes} threads: "
          f"{run_threads(lanes, batch8_points):8.1f} img/s")

    boxes8 = [api.Region(api.Point(256, 256), api.Point(768, 768))] * 8
    def batch8_boxes(_):
        segs = api.Segmentation.process_batch(views8, env)
        api.Segmentation.compute_mask_batch(segs, regions=boxes8)
        return 8
    print(f"{variant}: config 4 (batch 8, box prompts), ABI batch entry points, {lanes} threads: "
          f"{run_threads(lanes, batch8_boxes):8.1f} img/s")

    sizes = [(1800, 1200), (1024, 768), (512, 512), (640, 960), (1024, 1024)]
    mixed = [api.ImageView(synthetic_image(10 + i, width=w, height=h), api.Channels.rgba) for i, (w, h) in enumerate(sizes)]
    def mixed_5prompts(i):
        n = 0
        for v in mixed:
            seg = api.Segmentation.process(v, env)
            e = seg.extent()
            pts = [api.Point(int(e.width * fx), int(e.height * fy)) for fx, fy in ((.5, .5), (.25, .33), (.75, .2), (.6, .8), (.1, .9))]
            api.Segmentation.compute_mask_batch([seg] * 5, points=pts)
            n += 1
        return n
    r = run_threads(lanes, mixed_5prompts)
    print(f"{variant}: config 5 share (mixed resolution -> device resize, 5 prompts/image on the cached embedding), "
          f"{lanes} threads: {r:8.1f} img/s = {5 * r:8.1f} masks/s")

    # the same share as ONE batch per call, the form BASELINE config 5 names (16 images per GPU): slot 13 with 16 mixed-resolution
    # images, then slot 14 with their 80 prompts (bench.py's `configs.*.config5_share`)
    sizes16 = [sizes[i % len(sizes)] for i in range(16)]
    mixed16 = [api.ImageView(synthetic_image(10 + i, width=w, height=h), api.Channels.rgba) for i, (w, h) in enumerate(sizes16)]
    pts80 = [api.Point(int(w * fx), int(h * fy)) for (w, h) in sizes16 for fx, fy in ((.5, .5), (.25, .33), (.75, .2), (.6, .8), (.1, .9))]
    def mixed16_batch(_):
        segs = api.Segmentation.process_batch(mixed16, env)
        api.Segmentation.compute_mask_batch([sg for sg in segs for _ in range(5)], points=pts80)
        return 16
    for nt in sorted({1, 2, 3, lanes}):
        r = run_threads(nt, mixed16_batch)
        print(f"{variant}: config 5 share as batches (16 mixed-resolution images per call through slot 13, 80 prompts through slot 14), "
              f"{nt} thread(s): {r:8.1f} img/s = {5 * r:8.1f} masks/s")

    # the decode half of config 5 alone: prompts on embeddings that are already cached (the interactive use of the library)
    cached = [api.Segmentation.process(v, env) for v in mixed]
    def prompts_on_cached(i):
        seg = cached[i % len(cached)]
        e = seg.extent()
        pts = [api.Point(int(e.width * fx), int(e.height * fy)) for fx, fy in ((.5, .5), (.25, .33), (.75, .2), (.6, .8), (.1, .9))]
        api.Segmentation.compute_mask_batch([seg] * 5, points=pts)
        return 5
    print(f"{variant}: 5 prompts per call on cached embeddings (mixed sizes), 1 thread: {run_threads(1, prompts_on_cached):8.1f} masks/s, "
          f"{lanes} threads: {run_threads(lanes, prompts_on_cached):8.1f} masks/s")
    full = cached[-1]
    def one_prompt(_):
        full.compute_mask(api.Point(512, 512))
        return 1
    print(f"{variant}: 1 prompt per call on a cached 1024x1024 embedding, 1 thread: {run_threads(1, one_prompt):8.1f} masks/s, "
          f"{lanes} threads: {run_threads(lanes, one_prompt):8.1f} masks/s")
