"""End-to-end parity of the drop-in path (Segmentation.process / compute_mask through the C-ABI
table) against the CPU oracle on the same seeded weights, inputs and prompts."""
import numpy as np
import pytest

from conftest import (EMB_TOL, IOU_BAR, IOU_PRED_TOL, LOGIT_TOL, at_least, iou, single_mask_index, synthetic_image,
                      within)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from dlimgedit_amd import api
    return api


@pytest.fixture(scope="module")
def session(api, model_dirs):
    """(env, params, cfg, image, segmentation, oracle segmentation) for the reduced variant."""
    from oracle import sam_oracle as O
    mdir, params, cfg = model_dirs("vit_test")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    img = synthetic_image(0)
    seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
    ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
    return env, params, cfg, img, seg, ora


def test_backend_gate(api):
    assert api.Environment.is_supported(api.Backend.gpu)
    assert not api.Environment.is_supported(api.Backend.cpu)


def test_model_geometry(api, session):
    env, _, cfg, *_ = session
    assert api.ext.model_geometry(env) == (cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_dim)


def test_extent(api, session):
    seg = session[4]
    assert seg.extent() == api.Extent(1024, 1024)


def test_embedding_parity(api, session):
    *_, seg, ora = session
    emb = api.ext.get_embedding(seg)
    within("e2e.embedding", np.abs(emb - ora.embedding).max(), EMB_TOL)


@pytest.mark.parametrize("prompt", ["point", "region"])
def test_logits_parity(api, session, prompt):
    *_, seg, ora = session
    if prompt == "point":
        got, got_iou = api.ext.get_logits(seg, point=api.Point(512, 512))
        want, want_iou = ora.logits(point=(512, 512))
    else:
        r = api.Region(api.Point(256, 256), api.Point(768, 768))
        got, got_iou = api.ext.get_logits(seg, region=r)
        want, want_iou = ora.logits(region=(256, 256, 768, 768))
    within(f"e2e.logits.{prompt}", np.abs(got - want).max(), LOGIT_TOL)
    within(f"e2e.iou_pred.{prompt}", np.abs(got_iou - want_iou).max(), IOU_PRED_TOL)


def test_point_mask_iou(api, session):
    *_, seg, ora = session
    got = seg.compute_mask(api.Point(512, 512))
    want = ora.compute_mask(point=(512, 512))
    assert got.shape == (1024, 1024) and set(np.unique(got)) <= {0, 255}
    at_least("e2e.point_mask_iou", iou(got, want), IOU_BAR)


def test_region_mask_iou(api, session):
    *_, seg, ora = session
    got = seg.compute_mask(api.Region(api.Point(256, 256), api.Point(768, 768)))
    want = ora.compute_mask(region=(256, 256, 768, 768))
    at_least("e2e.region_mask_iou", iou(got, want), IOU_BAR)


def test_multi_mask_mode(api, session):
    """compute_masks: decoder outputs 1..3 with their predicted IoU (reference segmentation.cpp:167-172)."""
    *_, seg, ora = session
    got = seg.compute_masks(api.Point(300, 700))
    want_masks, want_acc = ora.compute_masks((300, 700))
    for t, (g, wm, wa) in enumerate(zip(got, want_masks, want_acc)):
        at_least(f"e2e.multi_mask_iou.{t}", iou(g.image, wm), IOU_BAR)
        within(f"e2e.multi_mask_accuracy.{t}", abs(g.accuracy - wa), IOU_PRED_TOL)


def test_mask_is_bit_exact_given_logits(api, session):
    """Post stage in isolation: feed the GPU's own logits to the oracle's post-processing."""
    from oracle import sam_oracle as O
    *_, seg, _ = session
    logits, iou_pred = api.ext.get_logits(seg, point=api.Point(512, 512))
    best = O.select_single(iou_pred, 2)
    want = O.write_mask_image(O.postprocess_logits(logits[best], (1024, 1024))[None, None], 0, (1024, 1024))
    got = seg.compute_mask(api.Point(512, 512))
    assert np.array_equal(got, want)


def test_repeated_queries_are_deterministic(api, session):
    *_, seg, _ = session
    a = seg.compute_mask(api.Point(100, 900))
    b = seg.compute_mask(api.Point(100, 900))
    assert np.array_equal(a, b)


def test_batch_entry_points(api, session):
    """process_images_for_segmentation / get_segmentation_masks equal the one-at-a-time calls."""
    env, *_ = session
    imgs = [synthetic_image(i) for i in (1, 2, 3)]
    views = [api.ImageView(i, api.Channels.rgba) for i in imgs]
    segs = api.Segmentation.process_batch(views, env)
    assert len(segs) == 3
    single = [api.Segmentation.process(v, env) for v in views]
    pts = [api.Point(512, 512), api.Point(200, 300), api.Point(900, 100)]
    batch_masks = api.Segmentation.compute_mask_batch(segs, points=pts)
    for sb, ss, p, mb in zip(segs, single, pts, batch_masks):
        assert np.array_equal(api.ext.get_embedding(sb), api.ext.get_embedding(ss))
        assert np.array_equal(mb, ss.compute_mask(p))


def test_channel_orders_end_to_end(api, session):
    """bgra / argb inputs that describe the same picture give the same embedding."""
    env, _, _, img, seg, _ = session
    ref = api.ext.get_embedding(seg)
    bgra = np.ascontiguousarray(img[:, :, [2, 1, 0, 3]])
    argb = np.ascontiguousarray(img[:, :, [3, 0, 1, 2]])
    for arr, ch in ((bgra, api.Channels.bgra), (argb, api.Channels.argb)):
        s = api.Segmentation.process(api.ImageView(arr, ch), env)
        assert np.array_equal(api.ext.get_embedding(s), ref)


def test_head_dim_80_variant(api, model_dirs):
    """ViT-H's head size (80) through the whole path on the reduced variant."""
    from oracle import sam_oracle as O
    mdir, params, cfg = model_dirs("vit_test80")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    img = synthetic_image(4)
    seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
    ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
    within("e2e.hd80.embedding", np.abs(api.ext.get_embedding(seg) - ora.embedding).max(), EMB_TOL)
    at_least("e2e.hd80.mask_iou", iou(seg.compute_mask(api.Point(512, 512)), ora.compute_mask(point=(512, 512))), IOU_BAR)


def test_folded_and_separate_layernorm_agree(api, session, model_dirs, monkeypatch):
    """The encoder's LayerNorms run inside the GEMMs by default; DLIMGEDIT_FUSED_LN=0 keeps them as separate kernels.
    Both must sit within the embedding tolerance of the oracle and close to each other."""
    env, params, cfg, img, seg, ora = session
    monkeypatch.setenv("DLIMGEDIT_FUSED_LN", "0")
    mdir, _, _ = model_dirs("vit_test")
    env_split = api.Environment(api.Options(api.Backend.gpu, mdir))
    seg_split = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env_split)
    folded, split = api.ext.get_embedding(seg), api.ext.get_embedding(seg_split)
    assert not np.array_equal(folded, split)                 # really two code paths
    within("e2e.ln_split.embedding", np.abs(split - ora.embedding).max(), EMB_TOL)
    within("e2e.ln_folded.embedding", np.abs(folded - ora.embedding).max(), EMB_TOL)
    within("e2e.ln_folded_vs_split", np.abs(folded - split).max(), EMB_TOL)
    at_least("e2e.ln_split.mask_iou", iou(seg_split.compute_mask(api.Point(512, 512)), ora.compute_mask(point=(512, 512))),
             IOU_BAR)


def test_default_options_consumer_runs_when_the_deployer_allows_it(api, session, model_dirs, monkeypatch):
    """A consumer that default-constructs Options (Backend::cpu, the reference's default) gets the same masks as a Backend::gpu
    consumer once the deployer has set DLIMGEDIT_CPU_REQUESTS_ON_GPU=1; without it the request is refused by name."""
    env, _, _, img, seg, _ = session
    mdir, _, _ = model_dirs("vit_test")
    with pytest.raises(api.Error, match="CPU backend is not available"):
        api.Environment(api.Options(api.Backend.cpu, mdir))
    monkeypatch.setenv("DLIMGEDIT_CPU_REQUESTS_ON_GPU", "1")
    cpu_env = api.Environment(api.Options(api.Backend.cpu, mdir))
    other = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), cpu_env)
    assert np.array_equal(api.ext.get_embedding(other), api.ext.get_embedding(seg))
    assert np.array_equal(other.compute_mask(api.Point(300, 400)), seg.compute_mask(api.Point(300, 400)))
    other.close()
    cpu_env.close()


def test_error_paths(api, session, tmp_path):
    env, *_ = session
    with pytest.raises(api.Error, match="does not exist"):
        api.Environment(api.Options(api.Backend.gpu, str(tmp_path / "nope")))
    with pytest.raises(api.Error, match="CPU backend"):
        api.Environment(api.Options(api.Backend.cpu, str(tmp_path)))
    empty = api.Environment(api.Options(api.Backend.gpu, str(tmp_path)))
    with pytest.raises(api.Error, match="Could not find model 'segmentation/sam_vit_b.dlw'"):
        api.Segmentation.process(api.ImageView(synthetic_image(0), api.Channels.rgba), empty)
    with pytest.raises(api.Error, match="not part of the MI355X build"):
        api.segment_objects(api.ImageView(synthetic_image(0), api.Channels.rgba), env)


def test_reference_wrapper_consumer_end_to_end(api, session, model_dirs, tmp_path):
    """A binary compiled ONLY against the reference's public C++ wrapper (oracle/_ref/abi_consumer, built in
    the build container by oracle/build_ref.py) runs Segmentation::process / compute_mask(s) on this library
    and gets the same mask as the Python binding."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = root / "oracle" / "_ref" / "abi_consumer"
    if not exe.exists():
        pytest.skip("oracle/_ref/abi_consumer was not built (reference tree absent at build time)")
    mdir, _, _ = model_dirs("vit_test")
    _, _, _, img, seg, _ = session
    raw = tmp_path / "img.raw"
    raw.write_bytes(img.tobytes())
    out = tmp_path / "mask.raw"
    r = subprocess.run([str(exe), str(root / "dlimgedit_amd" / "lib" / "libdlimgedit.so"), "run", mdir, str(raw),
                        "1024", "1024", "512", "512", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "extent=1024x1024" in r.stdout
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(1024, 1024)
    assert np.array_equal(got, seg.compute_mask(api.Point(512, 512)))
    acc = [float(v) for v in r.stdout.split("accuracy=")[1].split()]
    want = [m.accuracy for m in seg.compute_masks(api.Point(512, 512))]
    assert np.allclose(acc, want, atol=1e-5)


def test_reference_wrapper_consumer_loop_rate(model_dirs, tmp_path):
    """The rate an existing dlimgedit consumer sees: the same binary (reference wrapper headers only) runs its natural loop
    -- Segmentation::process(image), compute_mask(point), one thread, host buffers, the full-size ViT-B encoder -- for three
    seconds on this library, with the image held in a dlimg::Image (as after Image::load: memory the library allocated, which
    it pins, so the upload reads it in place and the result Image is written in place: csrc/image_memory.hpp) and with the
    image in the program's own buffer (staged).  No Python in the loop; the rates are left in gpurun_out/parity_margins.txt
    (the bar here is a sanity floor, not the target: VERDICT r05 item 5 asks for 480)."""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    exe = root / "oracle" / "_ref" / "abi_consumer"
    if not exe.exists():
        pytest.skip("oracle/_ref/abi_consumer was not built (reference tree absent at build time)")
    mdir, _, _ = model_dirs("vit_b")
    raw = tmp_path / "img.raw"
    raw.write_bytes(synthetic_image(0).tobytes())
    set_pixels = []
    for where, extra in (("pixels_in_an_Image", []), ("pixels_in_its_own_buffer", ["view"])):
        r = subprocess.run([str(exe), str(root / "dlimgedit_amd" / "lib" / "libdlimgedit.so"), "loop", mdir, str(raw),
                            "1024", "1024", "512", "512", "3"] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        fields = dict(f.split("=") for f in r.stdout.split())
        set_pixels.append(int(fields["set_pixels"]))
        at_least(f"e2e.reference_wrapper_consumer.images_per_s_one_thread.{where}", float(fields["images_per_s"]), 250)
    # the same mask whether the pixels were read in place (an Image is pinned memory of the library) or staged
    assert set_pixels[0] == set_pixels[1] > 0


def test_images_of_the_library_are_read_and_written_in_place(api, session, tmp_path):
    """The reference's wrapper allocates every Image through the table (create_image / load_image, dlimgedit.impl.hpp:139,
    165), so the pixels a consumer loaded and the Image compute_mask returns are memory of this library: pinned once a GPU
    environment exists (csrc/image_memory.hpp).  process() sends such pixels from where they lie -- and has finished reading
    them when it returns, although its encoder pass has not run yet -- and the post-processing kernel writes such a mask
    where the consumer reads it.  Same bits as from the program's own buffers, at 1024 x 1024, at a size that is resampled
    on the device, from a file, with rows that are not packed (staged), one mask and three."""
    env, _, _, _, _, _ = session
    for seed, (w, h) in ((3, (1024, 1024)), (4, (900, 640))):
        pixels = synthetic_image(seed, width=w, height=h)
        own = api.ImageView(pixels, api.Channels.rgba)
        held = api.Image(api.Extent(w, h), api.Channels.rgba)
        held.pixels()[...] = pixels
        assert api.ext.image_memory_is_pinned(held.pixels()) and not api.ext.image_memory_is_pinned(pixels)
        a = api.Segmentation.process(own, env)
        b = api.Segmentation.process(held.view(), env)
        held.pixels()[...] = 0                   # process() has returned: the pixels are the consumer's again
        pt = api.Point(w // 2, h // 2)
        ma, mb = a.compute_mask(pt), b.compute_mask(pt)
        assert api.ext.image_memory_is_pinned(mb) and np.array_equal(ma, mb) and ma.any()
        assert np.array_equal(api.ext.get_embedding(a), api.ext.get_embedding(b))
        plain = np.empty((h, w), dtype=np.uint8)           # a mask buffer of the program's own: staged and copied
        ptrs = (api.C.c_void_p * 3)(plain.ctypes.data, None, None)
        acc = (api.C.c_float * 3)()
        api._check(api.api().get_segmentation_mask(b._handle, (api.C.c_int * 2)(pt.x, pt.y), None, ptrs, acc))
        assert np.array_equal(plain, mb)
        three_a, three_b = a.compute_masks(pt), b.compute_masks(pt)
        assert all(np.array_equal(x.image, y.image) and x.accuracy == y.accuracy for x, y in zip(three_a, three_b))
        five = api.Segmentation.compute_mask_batch([b] * 5, points=[api.Point(10 + 100 * k, 20 + 90 * k) for k in range(5)])
        for k, m in enumerate(five):
            assert np.array_equal(m, a.compute_mask(api.Point(10 + 100 * k, 20 + 90 * k)))
        # rows that are not packed: a view into the middle of an Image is staged like any other strided view
        wide = api.Image(api.Extent(w + 16, h), api.Channels.rgba)
        wide.pixels()[:, :w] = pixels
        c = api.Segmentation.process(api.ImageView(wide.pixels()[:, :w], api.Channels.rgba, stride=(w + 16) * 4), env)
        assert np.array_equal(c.compute_mask(pt), ma)
        for sgm in (a, b, c):
            sgm.close()
    # from a file: load_image's pixels are image memory too
    path = tmp_path / "img.png"
    api.Image.save(api.ImageView(synthetic_image(5), api.Channels.rgba), path)
    loaded = api.Image.load(path)
    assert api.ext.image_memory_is_pinned(loaded.pixels())
    sa = api.Segmentation.process(loaded.view(), env)
    sb = api.Segmentation.process(api.ImageView(np.array(loaded.pixels()), api.Channels.rgba), env)
    assert np.array_equal(sa.compute_mask(api.Point(512, 512)), sb.compute_mask(api.Point(512, 512)))
    sa.close(), sb.close()
    # blocks go back to a free list by size and come out of it again; releasing twice or releasing foreign memory is ignored
    m1 = api.Image(api.Extent(1024, 1024), api.Channels.mask)
    addr = m1.pixels().ctypes.data
    del m1
    m2 = api.Image(api.Extent(1024, 1000), api.Channels.mask)        # same 64 KiB size class
    assert m2.pixels().ctypes.data == addr
    api.api().destroy_image(np.zeros(16, dtype=np.uint8).ctypes.data)


def test_pinned_image_memory_is_bounded(api, session):
    """Pinned pages cannot be swapped: beyond 4 GiB of live pinned image memory create_image hands out pageable blocks (which
    take the staged path), and of the released ones at most 512 MiB stay cached (csrc/image_memory.cpp)."""
    env, _, _, _, seg, _ = session
    big = api.Extent(8192, 8192)                 # 64 MiB per mask image
    held = [api.Image(big, api.Channels.mask) for _ in range(66)]
    pinned = [api.ext.image_memory_is_pinned(im.pixels()) for im in held]
    assert all(pinned[:60]) and not any(pinned[64:])
    held[-1].pixels()[:16, :16] = 7              # pageable blocks are ordinary memory
    del held
    small = api.Image(api.Extent(1024, 1024), api.Channels.mask)          # below the limit again: pinned
    assert api.ext.image_memory_is_pinned(small.pixels())
    assert np.array_equal(seg.compute_mask(api.Point(512, 512)), seg.compute_mask(api.Point(512, 512), out=np.empty((1024, 1024), np.uint8)))


@pytest.mark.parametrize("n_prompts", [2, 5, 6, 7])
def test_masks_written_straight_to_host_memory_equal_the_copied_ones(api, session, n_prompts):
    """Up to six masks of a call leave as one post-processing launch each, straight into pinned host memory (while the other
    lanes are idle; more go through a device buffer and piecewise copies: csrc/sam_model.cpp, enqueue_masks).  Either way the
    caller gets the masks of one-at-a-time queries, bit for bit -- on 1024 x 1024 and on a 1800 x 1200 image (2.1 MB masks at
    their own resolution, staging offsets padded to 256 bytes), single and multi-mask mode."""
    env, _, _, _, seg, _ = session
    wide = api.Segmentation.process(api.ImageView(synthetic_image(31, width=1800, height=1200, channels=3), api.Channels.rgb), env)
    for handle, (w, h) in ((seg, (1024, 1024)), (wide, (1800, 1200))):
        pts = [api.Point(37 + (w - 80) * k // n_prompts, h - 45 - (h - 90) * k // n_prompts) for k in range(n_prompts)]
        together = api.Segmentation.compute_mask_batch([handle] * n_prompts, points=pts)
        # the same into buffers of the program's own (staged and copied; the default results are Images of the library,
        # written in place: csrc/image_memory.hpp), and into a mix of both (one foreign buffer: all staged)
        own = [np.empty((h, w), dtype=np.uint8) for _ in pts]
        api.Segmentation.compute_mask_batch([handle] * n_prompts, points=pts, out=own)
        mixed = [api.Image(api.Extent(w, h), api.Channels.mask).pixels().reshape(h, w) for _ in pts]
        mixed[n_prompts // 2] = np.empty((h, w), dtype=np.uint8)
        api.Segmentation.compute_mask_batch([handle] * n_prompts, points=pts, out=mixed)
        for p, m, o, x in zip(pts, together, own, mixed):
            assert api.ext.image_memory_is_pinned(m) and not api.ext.image_memory_is_pinned(o)
            single = handle.compute_mask(p)
            assert m.shape == (h, w) and np.array_equal(m, single) and np.array_equal(o, single) and np.array_equal(x, single)
            assert np.array_equal(handle.compute_mask(p, out=np.empty((h, w), dtype=np.uint8)), single)
    three = wide.compute_masks(api.Point(900, 600))
    again = wide.compute_masks(api.Point(900, 600))
    assert all(np.array_equal(a.image, b.image) and a.accuracy == b.accuracy for a, b in zip(three, again))
    assert all(set(np.unique(m.image)) <= {0, 255} for m in three)
    wide.close()


@pytest.mark.parametrize("w,h,channels,point", [(1800, 1200, 3, (486, 722)), (512, 512, 4, (320, 210)),
                                                (640, 960, 4, (100, 800))])
def test_non_1024_images_end_to_end(api, session, w, h, channels, point):
    """Longest side != 1024: device resize -> padded pre-processing -> encode -> prompt in original
    coordinates (rounded as the reference does) -> crop + second bilinear back to the original size.
    Geometries of the reference's own integration tests (truck.jpg 1800x1200, cat_and_hat 512x512)."""
    from oracle import sam_oracle as O
    env, params, cfg, *_ = session
    img = synthetic_image(w, width=w, height=h, channels=4)[:, :, :channels].copy()
    ch = api.Channels.rgb if channels == 3 else api.Channels.rgba
    seg = api.Segmentation.process(api.ImageView(img, ch), env)
    assert seg.extent() == api.Extent(w, h)
    ora = O.OracleSegmentation(params, cfg).process(img, int(ch))
    within(f"e2e.{w}x{h}.embedding", np.abs(api.ext.get_embedding(seg) - ora.embedding).max(), EMB_TOL)
    got = seg.compute_mask(api.Point(*point))
    want = ora.compute_mask(point=point)
    assert got.shape == (h, w)
    at_least(f"e2e.{w}x{h}.mask_iou", iou(got, want), IOU_BAR)


@pytest.mark.parametrize("w,h", [(1, 1), (3, 1024), (2048, 16), (1024, 1), (17, 13), (4000, 3000)])
def test_extreme_geometries(api, session, w, h):
    """Degenerate and extreme aspect ratios go through resize, padding and the second bilinear without
    touching memory out of bounds; masks have the input's extent and match the oracle."""
    from oracle import sam_oracle as O
    env, params, cfg, *_ = session
    img = synthetic_image(w * 7 + h, width=w, height=h, channels=4)
    seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
    assert seg.extent() == api.Extent(w, h)
    pt = (w // 2, h // 2)
    got = seg.compute_mask(api.Point(*pt))
    assert got.shape == (h, w) and set(np.unique(got)) <= {0, 255}
    if w * h <= 2048 * 16:          # keep the oracle side cheap
        ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
        want = ora.compute_mask(point=pt)
        within(f"e2e.extreme.{w}x{h}.differing_pixels", (got != want).mean(), 0.002)


def test_prompts_outside_the_image_and_degenerate_boxes(api, session):
    *_, seg, ora = session
    for pt in [(-50, -50), (5000, 5000), (0, 0), (1023, 1023)]:
        got = seg.compute_mask(api.Point(*pt))
        want = ora.compute_mask(point=pt)
        at_least(f"e2e.outside.{pt[0]},{pt[1]}.mask_iou", iou(got, want), IOU_BAR)
    got = seg.compute_mask(api.Region(api.Point(700, 700), api.Point(100, 100)))      # inverted box
    want = ora.compute_mask(region=(700, 700, 100, 100))
    at_least("e2e.inverted_box.mask_iou", iou(got, want), IOU_BAR)


def test_mask_and_rgb_inputs(api, session):
    """1-channel (mask) and 3-channel inputs: the channel map replicates / selects as the reference does."""
    from oracle import sam_oracle as O
    env, params, cfg, img, *_ = session
    gray = np.ascontiguousarray(img[:, :, 0])
    seg = api.Segmentation.process(api.ImageView(gray, api.Channels.mask), env)
    ora = O.OracleSegmentation(params, cfg).process(gray, O.CH_MASK)
    within("e2e.mask_input.embedding", np.abs(api.ext.get_embedding(seg) - ora.embedding).max(), EMB_TOL)
    rgb = np.ascontiguousarray(img[:, :, :3])
    seg3 = api.Segmentation.process(api.ImageView(rgb, api.Channels.rgb), env)
    assert np.array_equal(api.ext.get_embedding(seg3), api.ext.get_embedding(session[4]))


def test_invalid_arguments_are_errors_not_crashes(api, session):
    env, _, _, img, seg, _ = session
    bad = api.ImageView(img, api.Channels.rgba)
    bad.channels = 2            # not a dlimg::Channels value
    with pytest.raises(api.Error, match="Unsupported channel order"):
        api.Segmentation.process(bad, env)
    narrow = api.ImageView(img, api.Channels.rgba)
    narrow.stride = 100         # smaller than one row
    with pytest.raises(api.Error, match="Assertion failed"):
        api.Segmentation.process(narrow, env)
    import ctypes as C
    masks = (C.c_void_p * 3)(None, None, None)
    acc = (C.c_float * 3)()
    pt = (C.c_int * 2)(1, 1)
    assert api.api().get_segmentation_mask(seg._handle, pt, None, masks, acc) == 1     # null output buffer
    assert b"Assertion failed" in api.api().last_error()
    assert api.api().get_segmentation_mask(seg._handle, None, None, masks, acc) == 1    # neither point nor region


@pytest.mark.parametrize("variant", ["vit_b", "vit_l", "vit_h"])
def test_full_size_models_against_committed_golden(api, model_dirs, monkeypatch, variant):
    """The real model sizes through the drop-in ABI against the committed fixtures (tests/golden/sam_<variant>.npz:
    samples of Hugging Face SamModel's embedding / low-res logits and its 1024x1024 masks on the same seeded weights
    and image).  No oracle in the loop."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / f"sam_{variant}.npz")
    mdir, _, _ = model_dirs(variant, int(g["seed"]))
    monkeypatch.setenv("DLIMGEDIT_SAM_MODEL", variant)
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    img = synthetic_image(int(g["image_seed"]))
    seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
    emb = api.ext.get_embedding(seg).reshape(-1)[::257]
    within(f"golden.{variant}.embedding", np.abs(emb - g["emb_samples"]).max(), EMB_TOL)
    for name, kw, call in (("point", dict(point=api.Point(512, 512)), lambda: seg.compute_mask(api.Point(512, 512))),
                           ("box", dict(region=api.Region(api.Point(256, 256), api.Point(768, 768))),
                            lambda: seg.compute_mask(api.Region(api.Point(256, 256), api.Point(768, 768))))):
        low, iou_pred = api.ext.get_logits(seg, **kw)
        within(f"golden.{variant}.{name}.logits", np.abs(low.reshape(4, -1)[:, ::61] - g[f"{name}_low_samples"]).max(), LOGIT_TOL)
        within(f"golden.{variant}.{name}.iou_pred", np.abs(iou_pred - g[f"{name}_iou"]).max(), IOU_PRED_TOL)
        # single-mask mode returns the best of tokens 1..3: the rule is applied to Hugging Face's own predictions
        best = single_mask_index(g[f"{name}_iou"])
        want = np.unpackbits(g[f"{name}_masks_bits"], axis=1).reshape(3, 1024, 1024)[best - 1] * 255
        at_least(f"golden.{variant}.{name}.mask_iou", iou(call(), want), IOU_BAR)


@pytest.mark.parametrize("variant", ["vit_test", "vit_b"])
def test_trained_like_activation_statistics(api, tmp_path_factory, monkeypatch, variant):
    """Weights with the activation statistics of trained ViTs (four "massive" residual channels at 150-250x the
    rest, LayerNorm scales over 0.05..5; dlimgedit_amd.weights.trained_like_weights) through the whole path: the
    folded-LayerNorm encoder (f16 copy of the RAW stream, mean correction in the consuming GEMM's epilogue) must stay
    inside the same tolerances as on the uniform synthetic weights, and so must the separate-LayerNorm build."""
    from dlimgedit_amd import weights as W
    from dlimgedit_amd.sam_config import get_config
    from oracle import sam_oracle as O
    cfg = get_config(variant)
    d = tmp_path_factory.mktemp(f"trained_like_{variant}")
    params = W.trained_like_weights(cfg, seed=3)
    W.save_weights(d / "segmentation" / W.weight_file_name(cfg), cfg, params)
    monkeypatch.setenv("DLIMGEDIT_SAM_MODEL", variant)
    img = synthetic_image(5)
    ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
    want = ora.compute_mask(point=(512, 512))
    errs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("DLIMGEDIT_FUSED_LN", fused)
        env = api.Environment(api.Options(api.Backend.gpu, str(d)))
        seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
        errs[fused] = float(np.abs(api.ext.get_embedding(seg) - ora.embedding).max())
        within(f"trained_like.{variant}.fused{fused}.embedding", errs[fused], EMB_TOL)
        at_least(f"trained_like.{variant}.fused{fused}.point_iou", iou(seg.compute_mask(api.Point(512, 512)), want), IOU_BAR)
        got_box = seg.compute_mask(api.Region(api.Point(256, 256), api.Point(768, 768)))
        at_least(f"trained_like.{variant}.fused{fused}.box_iou", iou(got_box, ora.compute_mask(region=(256, 256, 768, 768))),
                 IOU_BAR)
        seg.close()
        env.close()
    # folding the LayerNorm in must not cost more than a small multiple of the separate kernels' own f16 error
    assert errs["1"] <= 3.0 * errs["0"] + 5e-3, errs


def _hard_edged_image(seed: int, width: int = 1024, height: int = 1024) -> np.ndarray:
    """A second family of synthetic inputs (every other test uses low-frequency sinusoids + noise): flat-coloured rectangles
    and discs with hard edges on a gradient, saturated regions (0 and 255), a one-pixel checkerboard patch -- step
    edges, clipping and the highest spatial frequency a patch embedding can see."""
    rng = np.random.default_rng(5000 + seed)
    yy, xx = np.mgrid[0:height, 0:width]
    img = np.zeros((height, width, 4), np.uint8)
    for c in range(3):
        img[:, :, c] = (xx * (c + 1) * 255 // (3 * max(1, width - 1)) + yy * 40 // max(1, height - 1)).astype(np.uint8)
    for _ in range(14):
        x0, y0 = rng.integers(0, width - 40), rng.integers(0, height - 40)
        w, h = rng.integers(30, 400), rng.integers(30, 400)
        img[y0:y0 + h, x0:x0 + w, :3] = rng.integers(0, 256, 3)
    for _ in range(8):
        cx, cy, r = rng.integers(0, width), rng.integers(0, height), rng.integers(20, 180)
        img[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r, :3] = rng.integers(0, 256, 3)
    img[100:260, 700:900, :3] = 255
    img[800:960, 80:300, :3] = 0
    img[400:528, 400:528, :3] = (((xx[400:528, 400:528] + yy[400:528, 400:528]) & 1) * 255)[:, :, None]
    img[:, :, 3] = 255
    return img


@pytest.mark.parametrize("seed", [0, 1])
def test_hard_edged_images_against_the_oracle(api, session, seed):
    """Embedding, logits and masks on images with step edges, saturated areas and a one-pixel checkerboard (all other parity
    tests use smooth images): same tolerances."""
    from oracle import sam_oracle as O
    env, params, cfg, *_ = session
    img = _hard_edged_image(seed)
    seg = api.Segmentation.process(api.ImageView(img, api.Channels.rgba), env)
    ora = O.OracleSegmentation(params, cfg).process(img, O.CH_RGBA)
    within(f"e2e.hard_edges.{seed}.embedding", np.abs(api.ext.get_embedding(seg) - ora.embedding).max(), EMB_TOL)
    for name, gp, op in (("point", api.Point(450, 450), dict(point=(450, 450))), ("point2", api.Point(800, 180), dict(point=(800, 180))),
                         ("box", api.Region(api.Point(60, 700), api.Point(420, 1000)), dict(region=(60, 700, 420, 1000)))):
        got, got_iou = api.ext.get_logits(seg, **({"point": gp} if isinstance(gp, api.Point) else {"region": gp}))
        want, want_iou = ora.logits(**op)
        within(f"e2e.hard_edges.{seed}.{name}.logits", np.abs(got - want).max(), LOGIT_TOL)
        within(f"e2e.hard_edges.{seed}.{name}.iou_pred", np.abs(got_iou - want_iou).max(), IOU_PRED_TOL)
        got_mask, want_mask = seg.compute_mask(gp), ora.compute_mask(**op)
        within(f"e2e.hard_edges.{seed}.{name}.differing_pixels", (got_mask != want_mask).mean(), 0.002)
        if (want_mask > 0).mean() >= 0.02:          # IoU of a sliver says nothing
            at_least(f"e2e.hard_edges.{seed}.{name}.mask_iou", iou(got_mask, want_mask), IOU_BAR)
    seg.close()


def test_the_reference_s_truck_case_on_its_own_photograph(api, session):
    """The reference's integration test, restated (test/test_segmentation.cpp:137-140): Image::load("truck.jpg") ->
    Segmentation::process -> compute_mask(Point{486, 722}).  Its golden mask is a git-LFS stub and pins MobileSAM weights,
    so the comparison is with the oracle on the same decoded pixels and the same seeded weights -- but the INPUT is the
    reference's own photograph (tests/golden/truck.jpg, 1800 x 1200): a real image through the JPEG reader, the device
    resize, the encoder and the decoder, where every other test feeds synthetic pictures."""
    from pathlib import Path
    from oracle import sam_oracle as O
    env, params, cfg, *_ = session
    img = api.Image.load(Path(__file__).resolve().parent / "golden" / "truck.jpg")
    assert img.extent() == api.Extent(1800, 1200) and img.channels() == api.Channels.rgb
    pixels = img.pixels()
    seg = api.Segmentation.process(img.view(), env)
    assert seg.extent() == api.Extent(1800, 1200)
    ora = O.OracleSegmentation(params, cfg).process(np.ascontiguousarray(pixels), O.CH_RGB)
    within("e2e.truck.embedding", np.abs(api.ext.get_embedding(seg) - ora.embedding).max(), EMB_TOL)
    for name, gp, op in (("point", api.Point(486, 722), dict(point=(486, 722))),
                         ("box", api.Region(api.Point(180, 300), api.Point(1500, 1000)), dict(region=(180, 300, 1500, 1000)))):
        got, want = seg.compute_mask(gp), ora.compute_mask(**op)
        assert got.shape == (1200, 1800)
        within(f"e2e.truck.{name}.differing_pixels", (got != want).mean(), 0.002)
        if (want > 0).mean() >= 0.02:
            at_least(f"e2e.truck.{name}.mask_iou", iou(got, want), IOU_BAR)
    seg.close()


def test_values_beyond_the_f16_range_are_reported_not_decoded(api, model_dirs, tmp_path, monkeypatch):
    """No real checkpoint has run on this build, and its MFMA operands and residual stream are f16 (65504): a weight
    beyond that is refused by name when the model is loaded; an ACTIVATION beyond it -- here a positional-embedding channel
    at 1e5, far inside fp32 -- turns into an infinity somewhere in the encoder, and the handle says so instead of handing
    out masks that would be NaN patterns: process() of one image leaves the wait for its pass to the first call that needs
    the embedding (csrc/segmentation.hpp), so that call reports it, and every later one on the handle; a batch is waited
    for by process_batch() itself.  The environment keeps working afterwards."""
    from dlimgedit_amd import weights as W
    _, params, cfg = model_dirs("vit_test")
    img = api.ImageView(synthetic_image(0), api.Channels.rgba)

    heavy = dict(params)
    heavy["enc.L0.fc2.w"] = params["enc.L0.fc2.w"].copy()
    heavy["enc.L0.fc2.w"][3, 5] = 7.0e4
    d1 = tmp_path / "heavy"
    W.save_weights(d1 / "segmentation" / W.weight_file_name(cfg), cfg, heavy, allow_out_of_range=True)
    env = api.Environment(api.Options(api.Backend.gpu, str(d1)))
    with pytest.raises(api.Error, match=r"enc\.L0\.fc2\.w holds 70000.* outside the f16 range"):
        api.Segmentation.process(img, env)
    env.close()

    hot = dict(params)
    hot["enc.pos"] = params["enc.pos"].copy()
    hot["enc.pos"][:, 0] = 1.0e5
    d2 = tmp_path / "hot"
    W.save_weights(d2 / "segmentation" / W.weight_file_name(cfg), cfg, hot)
    env = api.Environment(api.Options(api.Backend.gpu, str(d2)))
    seg = api.Segmentation.process(img, env)
    for _ in range(2):
        with pytest.raises(api.Error, match="non-finite values: an activation left the f16 range"):
            seg.compute_mask(api.Point(512, 512))
    with pytest.raises(api.Error, match="non-finite values"):
        api.Segmentation.compute_mask_batch([seg], points=[api.Point(512, 512)])
    with pytest.raises(api.Error, match="non-finite values"):
        api.ext.get_embedding(seg)
    seg.close()
    api.Segmentation.process(img, env).close()          # never queried: nothing to report, nothing left behind
    monkeypatch.setenv("DLIMGEDIT_SYNC_PROCESS", "1")   # the deployer's switch: process() waits itself, as the reference's does
    with pytest.raises(api.Error, match="non-finite values: an activation left the f16 range"):
        api.Segmentation.process(img, env)
    monkeypatch.delenv("DLIMGEDIT_SYNC_PROCESS")
    with pytest.raises(api.Error, match="non-finite values"):
        api.Segmentation.process_batch([img, img, img], env)
    # a flag has an owner (r06): the reports above belonged to process() / process_batch() and were consumed there -- a
    # later synchronize of the asynchronous entry point, which queued nothing, has nothing to report
    api.ext.synchronize(env)
    # ... while a queued pass of its own is reported by the synchronize that retires it, once, with the request count
    dev_img = api.ext.device_alloc(env, 1024 * 1024 * 4)
    dev_mask = api.ext.device_alloc(env, 1024 * 1024)
    api.ext.copy_to_device(env, dev_img, synthetic_image(0))
    views = api.ext.device_views([dev_img], 1024, 1024)
    for _ in range(3):
        api.ext.encode_and_mask(env, views, [api.Point(512, 512)], [dev_mask])
    with pytest.raises(api.Error, match=r"3 request\(s\) ran in an image encoder pass that produced non-finite values"):
        api.ext.synchronize(env)
    api.ext.synchronize(env)
    api.ext.device_free(env, dev_img)
    api.ext.device_free(env, dev_mask)
    env.close()

    # the same images on the unmodified weights still work in this process (nothing sticky, no poisoned buffers)
    mdir, _, _ = model_dirs("vit_test")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    seg = api.Segmentation.process(img, env)
    assert np.isfinite(api.ext.get_embedding(seg)).all()
    seg.close()
    env.close()
