"""Hammer the decoder from several threads on ONE cached embedding and compare every result bit for bit with the serial
answer; on a mismatch say where the logits differ (plane, rows, how many, how much).
python tools/decoder_stress.py [threads] [reps per thread] [logits|masks|state|encode] [variant]"""
import sys, tempfile, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np
from conftest import synthetic_image
from dlimgedit_amd import api, weights as W
from dlimgedit_amd.sam_config import get_config

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "logits"      # logits | masks (compute_mask / compute_masks mixed, as the test does)
cfg = get_config(sys.argv[4] if len(sys.argv) > 4 else "vit_test")      # the decoder is the same for every variant; "encode" wants vit_b
with tempfile.TemporaryDirectory() as d:
    W.write_synthetic_model_dir(d, cfg, seed=0)
    env = api.Environment(api.Options(api.Backend.gpu, d))
    seg = api.Segmentation.process(api.ImageView(synthetic_image(41), api.Channels.rgba), env)
    prompts = [api.Point(150 + 90 * i, 900 - 85 * i) for i in range(8)]
    want = [api.ext.get_logits(seg, p) for p in prompts]
    # serial repeat first: every lane must agree with itself
    for rep in range(8):
        for j, p in enumerate(prompts):
            lg, iou = api.ext.get_logits(seg, p)
            assert np.array_equal(lg, want[j][0]) and np.array_equal(iou, want[j][1]), ("serial mismatch", rep, j)
    bad = []
    lock = threading.Lock()
    boxes = [api.Region(api.Point(40 * i, 30 * i), api.Point(600 + 50 * i, 500 + 60 * i)) for i in range(8)]
    want_pt = [seg.compute_mask(p) for p in prompts]
    want_box = [seg.compute_mask(b) for b in boxes]
    want_multi = [[m.image for m in seg.compute_masks(p)] for p in prompts[:3]]

    def describe(kind, t, rep, j, got, ref):
        diff = got != ref
        ys, xs = np.nonzero(diff)
        return dict(kind=kind, thread=t, rep=rep, prompt=j, pixels=int(diff.sum()), y=(int(ys.min()), int(ys.max())),
                    x=(int(xs.min()), int(xs.max())), ref_on=int((ref > 0).sum()), got_on=int((got > 0).sum()))

    def mask_worker(t):
        for rep in range(reps):
            for i in range(8):
                j = (i + 2 * t + rep) % 8
                g = seg.compute_mask(prompts[j])
                if not np.array_equal(g, want_pt[j]):
                    with lock: bad.append(describe("point", t, rep, j, g, want_pt[j]))
                g = seg.compute_mask(boxes[j])
                if not np.array_equal(g, want_box[j]):
                    with lock: bad.append(describe("box", t, rep, j, g, want_box[j]))
            got = seg.compute_masks(prompts[t % 3])
            for k, (g, w) in enumerate(zip(got, want_multi[t % 3])):
                if not np.array_equal(g.image, w):
                    with lock: bad.append(describe(f"multi{k}", t, rep, t % 3, g.image, w))

    def worker(t):
        for rep in range(reps):
            j = (rep + 2 * t) % 8
            lg, iou = api.ext.get_logits(seg, prompts[j])
            if not (np.array_equal(lg, want[j][0]) and np.array_equal(iou, want[j][1])):
                diff = lg != want[j][0]
                planes = [int(diff[m].sum()) for m in range(4)]
                ys, xs = np.nonzero(diff.any(axis=0))
                info = dict(thread=t, rep=rep, prompt=j, planes=planes, max_abs=float(np.abs(lg - want[j][0]).max()),
                            iou_diff=float(np.abs(iou - want[j][1]).max()),
                            y_range=(int(ys.min()), int(ys.max())) if len(ys) else None,
                            x_range=(int(xs.min()), int(xs.max())) if len(xs) else None)
                with lock:
                    bad.append(info)

    want_state = [api.ext.decoder_state(seg, p) for p in prompts] if mode == "state" else None
    images = [api.ImageView(synthetic_image(60 + i), api.Channels.rgba) for i in range(4)] if mode == "encode" else []
    want_emb = [api.ext.get_embedding(api.Segmentation.process(v, env)) for v in images]

    def encode_worker(t):
        for rep in range(reps):
            j = (rep + t) % len(images)
            emb = api.ext.get_embedding(api.Segmentation.process(images[j], env))
            d = emb != want_emb[j]
            if d.any():
                rows = np.nonzero(d.any(axis=1))[0]
                with lock: bad.append(dict(thread=t, rep=rep, image=j, elements=int(d.sum()), rows=(int(rows.min()), int(rows.max())),
                                           max_abs=float(np.abs(emb - want_emb[j]).max())))

    def state_worker(t):
        for rep in range(reps):
            j = (rep + 2 * t) % 8
            st = api.ext.decoder_state(seg, prompts[j])
            differing = []
            for n in st:
                d = st[n] != want_state[j][n]
                if d.any():
                    idx = np.nonzero(d)[0]
                    where = [int(i) for i in idx[:3]] if len(idx) <= 3 else None
                    differing.append((n, int(d.sum()), float(np.abs(st[n] - want_state[j][n]).max()), where,
                                      [float(st[n][i]) for i in idx[:2]] if len(idx) <= 3 else None,
                                      [float(want_state[j][n][i]) for i in idx[:2]] if len(idx) <= 3 else None))
            if differing:
                with lock: bad.append(dict(thread=t, rep=rep, prompt=j, differing=differing))

    ts = [threading.Thread(target={'masks': mask_worker, 'state': state_worker, 'encode': encode_worker}.get(mode, worker), args=(t,)) for t in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print(f"{threads} threads x {reps} decodes: {len(bad)} mismatches")
    for b in bad[:12]:
        print(b)
