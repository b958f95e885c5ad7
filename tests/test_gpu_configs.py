"""BASELINE.json configs 3, 4 and 5 at their full size through the drop-in ABI (batch slots 13-14), against the committed
Hugging Face fixtures (tests/golden/sam_vit_{b,h}*.npz, no oracle in the loop) and against the one-at-a-time calls
(bit-equal).  Prompt kinds and geometries follow the reference's integration tests (test/test_segmentation.cpp:101-150:
a point prompt, a region prompt, truck.jpg's 1800x1200).

  config 3   ViT-B, 8 images per GPU, one point prompt each
  config 4   ViT-H, 8 images, region (box) prompts
  config 5   ViT-H, mixed resolutions {1800x1200, 1024x768, 512x512, 640x960, 1024x1024} resized to 1024 on the device,
             5 Halton point prompts per image on the cached embedding
"""
import os
from pathlib import Path

import numpy as np
import pytest

from conftest import (EMB_TOL, IOU_BAR, IOU_PRED_TOL, LOGIT_TOL, at_least, halton_points, iou, single_mask_index,
                      synthetic_image, within)

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
EMB_STRIDE, LOW_STRIDE = 257, 61


@pytest.fixture(scope="module")
def api():
    from dlimgedit_amd import api
    return api


@pytest.fixture(scope="module")
def full_env(api, model_dirs):
    """variant -> environment on the seeded full-size weights of the fixtures (kept for the module: ViT-H is 2.5 GB)."""
    envs = {}

    def get(variant):
        if variant not in envs:
            g = np.load(GOLD / f"sam_{variant}.npz")
            mdir, _, _ = model_dirs(variant, int(g["seed"]))
            saved = os.environ.get("DLIMGEDIT_SAM_MODEL")
            os.environ["DLIMGEDIT_SAM_MODEL"] = variant
            try:
                env = api.Environment(api.Options(api.Backend.gpu, mdir))
                api.ext.model_geometry(env)      # loads the weights while the variable names this variant
            finally:
                if saved is None:
                    os.environ.pop("DLIMGEDIT_SAM_MODEL", None)
                else:
                    os.environ["DLIMGEDIT_SAM_MODEL"] = saved
            envs[variant] = env
        return envs[variant]

    yield get
    for e in envs.values():
        e.close()


def _golden_masks(g, name):
    return np.unpackbits(g[f"{name}_masks_bits"], axis=1).reshape(3, 1024, 1024) * 255


def _batch_of_eight(api, env, g, prompt_kind):
    """Eight images through process_images_for_segmentation + get_segmentation_masks; image 0 is the fixture's."""
    imgs = [synthetic_image(int(g["image_seed"]) + i) for i in range(8)]
    views = [api.ImageView(im, api.Channels.rgba) for im in imgs]
    segs = api.Segmentation.process_batch(views, env)
    assert len(segs) == 8
    if prompt_kind == "point":
        prompt = api.Point(512, 512)
        masks = api.Segmentation.compute_mask_batch(segs, points=[prompt] * 8)
    else:
        prompt = api.Region(api.Point(256, 256), api.Point(768, 768))
        masks = api.Segmentation.compute_mask_batch(segs, regions=[prompt] * 8)
    # image 0 against Hugging Face
    emb = api.ext.get_embedding(segs[0])
    within(f"config.batch8.{prompt_kind}.embedding", np.abs(emb.reshape(-1)[::EMB_STRIDE] - g["emb_samples"]).max(), EMB_TOL)
    name = "point" if prompt_kind == "point" else "box"
    best = single_mask_index(g[f"{name}_iou"])
    at_least(f"config.batch8.{prompt_kind}.mask_iou", iou(masks[0], _golden_masks(g, name)[best - 1]), IOU_BAR)
    # every image: the batch slots give exactly what the one-at-a-time slots give
    for view, seg, mask in zip(views, segs, masks):
        assert mask.shape == (1024, 1024) and set(np.unique(mask)) <= {0, 255}
        one = api.Segmentation.process(view, env)
        assert np.array_equal(api.ext.get_embedding(one), api.ext.get_embedding(seg))
        assert np.array_equal(one.compute_mask(prompt), mask)
        one.close()
    # the eight images are different pictures: so are their masks
    assert len({m.tobytes() for m in masks}) == 8
    for s in segs:
        s.close()


def test_config3_vit_b_batch8_point_prompts(api, full_env):
    _batch_of_eight(api, full_env("vit_b"), np.load(GOLD / "sam_vit_b.npz"), "point")


def test_config4_vit_h_batch8_region_prompts(api, full_env):
    _batch_of_eight(api, full_env("vit_h"), np.load(GOLD / "sam_vit_h.npz"), "region")


def test_multi_mask_mode_full_size(api, full_env):
    """compute_masks at full size: decoder outputs 1..3 and their IoU predictions against Hugging Face."""
    g = np.load(GOLD / "sam_vit_b.npz")
    env = full_env("vit_b")
    seg = api.Segmentation.process(api.ImageView(synthetic_image(int(g["image_seed"])), api.Channels.rgba), env)
    got = seg.compute_masks(api.Point(512, 512))
    want = _golden_masks(g, "point")
    for t in range(3):
        at_least(f"config.multi_mask.{t}.iou", iou(got[t].image, want[t]), IOU_BAR)
        within(f"config.multi_mask.{t}.accuracy", abs(got[t].accuracy - float(g["point_iou"][t + 1])), IOU_PRED_TOL)
    seg.close()


CONFIG5_SIZES = [(1800, 1200), (1024, 768), (512, 512), (640, 960), (1024, 1024)]


def test_config5_vit_h_mixed_resolution_five_prompts_on_cached_embedding(api, full_env):
    """BASELINE config 5 at one GPU's share: 16 images (the five sizes cycled), five Halton point
    prompts per cached embedding, all 80 prompts in ONE slot-14 call."""
    env = full_env("vit_h")
    g = np.load(GOLD / "sam_vit_h_1800x1200.npz")
    sizes = [CONFIG5_SIZES[i % len(CONFIG5_SIZES)] for i in range(16)]
    imgs, views = [], []
    for i, (w, h) in enumerate(sizes):
        # image 0 is the fixture's picture (RGB like the reference's truck.jpg); later repeats of a size are new pictures
        im = synthetic_image(w + 31 * (i // len(CONFIG5_SIZES)), width=w, height=h, channels=4)
        if i == 0:
            im = im[:, :, :3].copy()
            views.append(api.ImageView(im, api.Channels.rgb))
        else:
            views.append(api.ImageView(im, api.Channels.rgba))
        imgs.append(im)
    segs = api.Segmentation.process_batch(views, env)
    assert len(segs) == 16
    prompts = [halton_points(5, w, h, start=1 + 5 * (i // len(CONFIG5_SIZES))) for i, (w, h) in enumerate(sizes)]
    assert prompts[0] == [tuple(p) for p in g["points"].tolist()]
    # 80 prompts in ONE call: every image's embedding is used five times
    flat_segs = [s for s in segs for _ in range(5)]
    flat_pts = [api.Point(*p) for ps in prompts for p in ps]
    masks = api.Segmentation.compute_mask_batch(flat_segs, points=flat_pts)
    assert len(masks) == 80
    # ... into Images of the library (copied from the device to where they lie); into buffers of the caller's own the masks
    # pass through the pinned staging pieces: the same bits
    own = [np.empty(m.shape, dtype=np.uint8) for m in masks]
    api.Segmentation.compute_mask_batch(flat_segs, points=flat_pts, out=own)
    assert all(np.array_equal(a, b) for a, b in zip(own, masks))
    for k, (seg, pt, mask) in enumerate(zip(flat_segs, flat_pts, masks)):
        w, h = sizes[k // 5]
        assert seg.extent() == api.Extent(w, h)
        assert mask.shape == (h, w) and set(np.unique(mask)) <= {0, 255}
        if k % 5 in (0, 3) or k < 25:
            assert np.array_equal(seg.compute_mask(pt), mask)      # batch slot == single slot, bit for bit
    # images processed in one batch call == the same images one at a time (sizes 2..4 of the second cycle)
    for i in (6, 7, 8):
        one = api.Segmentation.process(views[i], env)
        assert np.array_equal(api.ext.get_embedding(one), api.ext.get_embedding(segs[i]))
        one.close()
    # 1800x1200 against Hugging Face (resize -> pad -> encode -> rounded prompt -> crop + second bilinear)
    emb = api.ext.get_embedding(segs[0])
    within("config5.1800x1200.embedding", np.abs(emb.reshape(-1)[::EMB_STRIDE] - g["emb_samples"]).max(), EMB_TOL)
    want_masks = np.unpackbits(g["masks_bits"], axis=2)[:, :, :1800 * 1200].reshape(5, 3, 1200, 1800) * 255
    for j in range(5):
        low, iou_pred = api.ext.get_logits(segs[0], point=flat_pts[j])
        within(f"config5.1800x1200.logits.{j}", np.abs(low.reshape(4, -1)[:, ::LOW_STRIDE] - g["low_samples"][j]).max(), LOGIT_TOL)
        within(f"config5.1800x1200.iou_pred.{j}", np.abs(iou_pred - g["iou"][j]).max(), IOU_PRED_TOL)
        best = single_mask_index(g["iou"][j])
        at_least(f"config5.1800x1200.mask_iou.{j}", iou(masks[j], want_masks[j, best - 1]), IOU_BAR)
    for s in segs:
        s.close()


def test_replicas_deal_images_round_robin_and_agree(api, model_dirs, monkeypatch):
    """DLIMGEDIT_DEVICES lists the GPUs an environment uses; the same GPU twice gives two independent replicas, which
    exercises the multi-device paths (per-replica host threads, handles pinned to the replica that holds their
    embedding, mask queries routed by handle) on a one-GPU box.  Results do not depend on the replica."""
    mdir, _, _ = model_dirs("vit_test")
    one = api.Environment(api.Options(api.Backend.gpu, mdir))
    monkeypatch.setenv("DLIMGEDIT_DEVICES", "0,0")
    two = api.Environment(api.Options(api.Backend.gpu, mdir))
    monkeypatch.delenv("DLIMGEDIT_DEVICES")
    assert api.ext.replica_count(one) == 1 and api.ext.replica_count(two) == 2
    imgs = [synthetic_image(20 + i) for i in range(5)]
    views = [api.ImageView(im, api.Channels.rgba) for im in imgs]
    segs1 = api.Segmentation.process_batch(views, one)
    segs2 = api.Segmentation.process_batch(views, two)
    assert [api.ext.segmentation_device(s) for s in segs1] == [(0, 0)] * 5
    placement = [api.ext.segmentation_device(s)[0] for s in segs2]
    assert sorted(placement) == [0, 0, 0, 1, 1] or sorted(placement) == [0, 0, 1, 1, 1]
    assert all(placement[i] != placement[i + 1] for i in range(4))       # image i -> replica i mod G
    pts = [api.Point(100 + 150 * i, 900 - 120 * i) for i in range(5)]
    m1 = api.Segmentation.compute_mask_batch(segs1, points=pts)
    m2 = api.Segmentation.compute_mask_batch(segs2, points=pts)
    for a, b, s1, s2, p in zip(m1, m2, segs1, segs2, pts):
        assert np.array_equal(api.ext.get_embedding(s1), api.ext.get_embedding(s2))
        assert np.array_equal(a, b)
        assert np.array_equal(s2.compute_mask(p), b)
    multi = segs2[1].compute_masks(pts[1])
    ref = segs1[1].compute_masks(pts[1])
    for x, y in zip(multi, ref):
        assert np.array_equal(x.image, y.image) and x.accuracy == y.accuracy
    with pytest.raises(api.Error, match="out of range"):
        monkeypatch.setenv("DLIMGEDIT_DEVICES", "0,99")
        api.Environment(api.Options(api.Backend.gpu, mdir))


def test_resize_table_cache_survives_more_sizes_than_it_holds(api, model_dirs):
    """The device-side resize keeps one contributor table per (input, output) extent of an axis, 64 per lane (LRU).  A
    host that cycles through more sizes while one axis repeats must keep getting the same pixels (the tables of the
    repeated axis are in use while the other axis' lookup evicts)."""
    mdir, _, _ = model_dirs("vit_test")
    os.environ["DLIMGEDIT_LANES"] = "1"
    try:
        env = api.Environment(api.Options(api.Backend.gpu, mdir))
        h = 1000                                  # longest side: scale and the (h -> 1024) table are the same every time
        first_img = synthetic_image(1, width=300, height=h)
        first = api.Segmentation.process(api.ImageView(first_img, api.Channels.rgba), env)
        want = api.ext.get_embedding(first)
        for i in range(70):                       # 70 further (w -> rw) tables go through a cache of 64
            w = 310 + 7 * i
            api.Segmentation.process(api.ImageView(synthetic_image(2, width=w, height=h), api.Channels.rgba), env).close()
        again = api.Segmentation.process(api.ImageView(first_img, api.Channels.rgba), env)
        assert np.array_equal(api.ext.get_embedding(again), want)
        env.close()
    finally:
        os.environ.pop("DLIMGEDIT_LANES", None)


def test_device_output_gather_matches_host_masks(api, model_dirs, monkeypatch):
    """SURVEY.md section 8e, the optional "all masks on one device" result: dlimg_amd_get_segmentation_masks_device
    produces every mask on the GPU that holds its embedding and delivers it into ONE buffer on the root device (peer
    copies over xGMI between GPUs, no host memory).  On the one-GPU box the environment lists GPU 0 twice (two
    replicas); the peer-copy branch is forced once through DLIMGEDIT_FORCE_PEER_COPY (same code, the copy degenerates to
    device-to-device).  Bit-equal to the host path of slot 14 either way, mixed image sizes included."""
    mdir, _, _ = model_dirs("vit_test")
    monkeypatch.setenv("DLIMGEDIT_DEVICES", "0,0")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    monkeypatch.delenv("DLIMGEDIT_DEVICES")
    sizes = [(1024, 1024), (640, 960), (1024, 768), (512, 512), (1024, 1024)]
    imgs = [synthetic_image(40 + i, width=w, height=h) for i, (w, h) in enumerate(sizes)]
    segs = api.Segmentation.process_batch([api.ImageView(im, api.Channels.rgba) for im in imgs], env)
    assert sorted(api.ext.segmentation_device(s)[0] for s in segs) in ([0, 0, 0, 1, 1], [0, 0, 1, 1, 1])
    # 11 prompts over 5 embeddings (more than one chunk of 8 on a replica is not needed: 2 replicas share them)
    order = [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0]
    pts = [api.Point(*halton_points(1, *sizes[i], start=3 + k)[0]) for k, i in enumerate(order)]
    many = [segs[i] for i in order]
    want = api.Segmentation.compute_mask_batch(many, points=pts)
    total = sum(w * h for (w, h) in (sizes[i] for i in order))
    dev = api.ext.device_alloc(env, total)
    try:
        for forced in ("0", "1"):
            monkeypatch.setenv("DLIMGEDIT_FORCE_PEER_COPY", forced)
            api.ext.copy_to_device(env, dev, np.full(total, 7, np.uint8))
            offsets = api.ext.compute_mask_batch_device(many, dev, points=pts, root_device=0)
            got = np.empty(total, np.uint8)
            api.ext.copy_to_host(env, got, dev)
            assert offsets[0] == 0 and all(offsets[k + 1] - offsets[k] == sizes[order[k]][0] * sizes[order[k]][1]
                                           for k in range(len(order) - 1))
            for k, i in enumerate(order):
                w, h = sizes[i]
                assert np.array_equal(got[offsets[k]:offsets[k] + w * h].reshape(h, w), want[k]), (forced, k)
        monkeypatch.delenv("DLIMGEDIT_FORCE_PEER_COPY")
        # regions and the error path
        regs = [api.Region(api.Point(10, 20), api.Point(sizes[i][0] - 30, sizes[i][1] - 40)) for i in order[:3]]
        want_r = api.Segmentation.compute_mask_batch(many[:3], regions=regs)
        offsets = api.ext.compute_mask_batch_device(many[:3], dev, regions=regs, root_device=0)
        got = np.empty(total, np.uint8)
        api.ext.copy_to_host(env, got, dev)
        for k in range(3):
            w, h = sizes[order[k]]
            assert np.array_equal(got[offsets[k]:offsets[k] + w * h].reshape(h, w), want_r[k])
        with pytest.raises(api.Error, match="out of range"):
            api.ext.compute_mask_batch_device(many[:1], dev, points=pts[:1], root_device=99)
    finally:
        api.ext.device_free(env, dev)
    for s in segs:
        s.close()
    env.close()


def test_device_output_gather_config5_eighty_prompts(api, model_dirs, monkeypatch):
    """BASELINE config 5 in the shape one rank of the 8-GPU job sees it (16 mixed-resolution images, five Halton prompts per
    cached embedding) through the device-output gather: ONE call with 80 prompts, two replicas (GPU 0 listed twice) and
    the peer-copy branch forced, so every mask takes the hipMemcpyPeerAsync route into the root's buffer.  Bit-equal to
    slot 14's host masks; offsets tightly packed in prompt order."""
    mdir, _, _ = model_dirs("vit_test")
    monkeypatch.setenv("DLIMGEDIT_DEVICES", "0,0")
    monkeypatch.setenv("DLIMGEDIT_FORCE_PEER_COPY", "1")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    monkeypatch.delenv("DLIMGEDIT_DEVICES")
    base = [(1800, 1200), (1024, 768), (512, 512), (640, 960), (1024, 1024)]
    sizes = [base[i % 5] for i in range(16)]
    views = [api.ImageView(synthetic_image(70 + i, width=w, height=h), api.Channels.rgba) for i, (w, h) in enumerate(sizes)]
    segs = api.Segmentation.process_batch(views, env)
    placement = [api.ext.segmentation_device(s)[0] for s in segs]
    assert placement.count(0) == 8 and placement.count(1) == 8
    many, pts, extents = [], [], []
    for i, (w, h) in enumerate(sizes):
        for x, y in halton_points(5, w, h, start=1 + i):
            many.append(segs[i])
            pts.append(api.Point(x, y))
            extents.append((w, h))
    assert len(many) == 80
    want = api.Segmentation.compute_mask_batch(many, points=pts)
    total = sum(w * h for w, h in extents)
    dev = api.ext.device_alloc(env, total)
    try:
        api.ext.copy_to_device(env, dev, np.full(total, 7, np.uint8))
        offsets = api.ext.compute_mask_batch_device(many, dev, points=pts, root_device=0)
        got = np.empty(total, np.uint8)
        api.ext.copy_to_host(env, got, dev)
        assert offsets[0] == 0
        assert all(offsets[k + 1] - offsets[k] == extents[k][0] * extents[k][1] for k in range(79))
        for k, (w, h) in enumerate(extents):
            m = got[offsets[k]:offsets[k] + w * h].reshape(h, w)
            assert np.array_equal(m, want[k]), k
            assert set(np.unique(m)) <= {0, 255}
    finally:
        api.ext.device_free(env, dev)
    for s in segs:
        s.close()
    env.close()


def test_batched_identity_postprocessing_on_ragged_extents(api, model_dirs):
    """Launches of four or more masks whose second resize stage is the identity (longest side 1024) take the 16 x 4-pixels-per-thread
    form of the post-processing kernel (kernels/postprocess.hip); one mask at a time takes the general per-pixel kernel.
    Heights / widths that are not multiples of 16 or of 4 exercise its ragged bottom and right edges: bit-equal either way."""
    mdir, _, _ = model_dirs("vit_test")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    sizes = [(1024, 700), (600, 1024), (1024, 1022), (1024, 1024), (1022, 1024), (1024, 52)]
    views = [api.ImageView(synthetic_image(90 + i, width=w, height=h), api.Channels.rgba) for i, (w, h) in enumerate(sizes)]
    segs = api.Segmentation.process_batch(views, env)
    pts = [api.Point(w // 2, h // 2) for (w, h) in sizes]
    batch = api.Segmentation.compute_mask_batch(segs, points=pts)
    for seg, p, got, (w, h) in zip(segs, pts, batch, sizes):
        assert got.shape == (h, w)
        assert np.array_equal(seg.compute_mask(p), got), (w, h)
    for s in segs:
        s.close()
    env.close()
