"""The reference's own known-answer tests for the host glue of the SAM path, restated against the
CPU oracle (reference: test/test_segmentation.cpp:15-99).  These pin the oracle's glue functions."""
import numpy as np
import pytest

from oracle import sam_oracle as O


@pytest.mark.parametrize("w,h,max_side,expect", [
    (13, 19, 26, (18, 26)), (13, 19, 10, (7, 10)),      # "height" sections
    (19, 13, 26, (26, 18)), (19, 13, 10, (10, 7)),      # "width" sections
])
def test_resize_longest_side_extent(w, h, max_side, expect):
    """ResizeLongestSide.resize (test_segmentation.cpp:15-46): target = int(dim*scale + 0.5)."""
    assert O.ResizeLongestSide(max_side).target_extent(w, h) == expect


def test_resize_longest_side_identity_at_1024():
    rs = O.ResizeLongestSide(1024)
    assert rs.target_extent(1024, 683) == (1024, 683) and rs.scale == 1


def test_resize_longest_side_transform():
    """test_segmentation.cpp:48-57."""
    rs = O.ResizeLongestSide(20)
    assert rs.target_extent(10, 10) == (20, 20)
    assert rs.transform(0, 0) == (0, 0)
    assert rs.transform(10, 10) == (20, 20)
    assert rs.transform(2, 7) == (4, 14)


@pytest.mark.parametrize("channels,nbytes,expected", [
    (O.CH_RGBA, 4, (0, 1, 2, 4, 5, 32)),
    (O.CH_RGB, 3, (0, 1, 2, 3, 4, 24)),
    (O.CH_BGRA, 4, (2, 1, 0, 6, 5, 34)),
    (O.CH_ARGB, 4, (1, 2, 3, 5, 6, 33)),
])
def test_create_image_tensor(channels, nbytes, expected):
    """SAM.create_image_tensor (test_segmentation.cpp:59-83): 8x6 iota image."""
    img = np.arange(8 * 6 * nbytes, dtype=np.uint8).reshape(6, 8, nbytes)
    t = O.create_image_tensor(img, channels)
    assert t.dtype == np.float32 and t.shape == (6, 8, 3)
    got = (t[0, 0, 0], t[0, 0, 1], t[0, 0, 2], t[0, 1, 0], t[0, 1, 1], t[1, 0, 0])
    assert got == tuple(float(e) for e in expected)


def test_create_image_tensor_mask_replicates():
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    t = O.create_image_tensor(img, O.CH_MASK)
    assert np.array_equal(t[:, :, 0], t[:, :, 1]) and np.array_equal(t[:, :, 0], t[:, :, 2])


def test_write_mask_image():
    """SAM.write_mask_image (test_segmentation.cpp:85-99): strict > 0, cropped with the tensor's stride."""
    vals = np.array([0.0, 0.0, 0.2, -3.1, 0.0, 5.5, 0.0, 0.7, 0.0, 0.9], dtype=np.float32).reshape(1, 1, 2, 5)
    m = O.write_mask_image(vals, 0, (4, 2))
    assert m.tolist() == [[0, 0, 255, 0], [255, 0, 255, 0]]


def test_pack_prompt_point_and_region():
    """segmentation.cpp:135-152: point -> labels (1,-1) with the (0,0) pad point; region -> labels (2,3)."""
    rs = O.ResizeLongestSide(1024)
    rs.target_extent(512, 512)          # scale 2
    c, l = O.pack_prompt(rs, point=(320, 210))
    assert c.tolist() == [[640, 420], [0, 0]] and l.tolist() == [1, -1]
    c, l = O.pack_prompt(rs, region=(180, 110, 505, 330))
    assert c.tolist() == [[360, 220], [1010, 660]] and l.tolist() == [2, 3]


def test_coordinate_rounding_for_non_integer_scale():
    rs = O.ResizeLongestSide(1024)
    assert rs.target_extent(1800, 1200) == (1024, 683)
    # int(486 * 0.5688889 + 0.5) etc.: coordinates are rounded to integers in the resized image
    assert rs.transform(486, 722) == (276, 411)


def test_select_single_mask_rule():
    """SamOnnxModel.select_masks with 2 prompt tokens: token 0 gets -500 and loses unless far ahead."""
    assert O.select_single(np.array([0.9, 0.1, 0.5, 0.3], np.float32), 2) == 2
    assert O.select_single(np.array([600.0, 0.1, 0.5, 0.3], np.float32), 2) == 0
    assert O.select_single(np.array([0.2, 0.7, 0.7, 0.1], np.float32), 2) == 1     # first maximum wins


def test_preprocess_pads_with_zero_in_normalised_space():
    img = np.full((2, 3, 3), 255, np.float32)
    x = O.preprocess(img)
    assert x.shape == (3, 1024, 1024)
    assert np.all(x[:, 2:, :] == 0) and np.all(x[:, :, 3:] == 0)
    assert np.allclose(x[:, 0, 0], (255 - O.PIXEL_MEAN) / O.PIXEL_STD)


def test_patchify_order():
    chw = np.arange(3 * 32 * 32, dtype=np.float32).reshape(3, 32, 32)
    p = O.patchify(chw, 16)
    assert p.shape == (4, 768)
    # row 1 = patch (py 0, px 1); column c*256 + iy*16 + ix
    assert p[1, 1 * 256 + 2 * 16 + 3] == chw[1, 2, 16 + 3]
    assert p[2, 0] == chw[0, 16, 0]


def test_image_resize_kat():
    """Image resize (reference test/test_image.cpp:51-69): 8x8 RGBA ramp -> 4x4 through the restated
    stb_image_resize call (Mitchell, sRGB, clamp): R = A = 255, G = 2 + 8*row, B = 2 + 8*col."""
    from oracle import stb_resize as R
    img = np.zeros((8, 8, 4), np.uint8)
    for i in range(64):
        img[i // 8, i % 8] = (255, 4 * (i // 8), 4 * (i % 8), 255)
    out = R.resize_srgb(img, 4, 4)
    assert out.shape == (4, 4, 4)
    for i in range(16):
        assert tuple(int(v) for v in out[i // 4, i % 4]) == (255, 2 + 8 * (i // 4), 2 + 8 * (i % 4), 255)


def _check_against_pillow_upsample(resize):
    """resize(src, out_w, out_h) against tests/golden/resize_upsample_pil.npz (Pillow BICUBIC = Catmull-Rom, a = -0.5, in
    linear light with the sRGB formulas in float64; made by tests/golden/make_resize_golden.py, no oracle code in it).
    Not bit-exact by construction (border treatment, float tables): every byte within 1 LSB, at most 3 % of them off."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "resize_upsample_pil.npz")
    for name in ("rgb_96x60_to_256x160", "rgba_50x80_to_160x256"):
        src, want = g[name + "_src"], g[name + "_out"]
        got = resize(src, want.shape[1], want.shape[0])
        assert got.shape == want.shape and got.dtype == np.uint8
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1, (name, int(d.max()))
        assert (d > 0).mean() < 0.03, (name, float((d > 0).mean()))


def test_image_resize_upsampling_against_pillow_catmull_rom():
    """The up-sampling half of the default filter (Catmull-Rom), which the reference's KAT does not reach."""
    from oracle import stb_resize as R
    _check_against_pillow_upsample(R.resize_srgb)


def test_srgb_tables_round_trip_and_are_monotonic():
    """Giesen's float->sRGB8 conversion must invert the 256-entry decode table and never decrease."""
    from oracle import stb_resize as R
    t = R.srgb_to_linear_table()
    assert np.array_equal(R.linear_to_srgb_uchar(t), np.arange(256, dtype=np.uint8))
    y = R.linear_to_srgb_uchar(np.linspace(0, 1, 200001, dtype=np.float32)).astype(int)
    assert np.all(np.diff(y) >= 0) and y[0] == 0 and y[-1] == 255
    assert R.linear_to_srgb_uchar(np.array([np.nan, -1.0, 2.0], np.float32)).tolist() == [0, 0, 255]


def test_resize_contributors_sum_to_one():
    from oracle import stb_resize as R
    for n_in, n_out in [(8, 4), (1800, 1024), (512, 1024), (1200, 683), (13, 18), (19, 26), (5, 5)]:
        first, count, coef = R.axis_contributors(n_in, n_out)
        assert np.allclose(coef.sum(axis=1), 1.0, atol=1e-5)
        assert count.min() >= 1 and first.min() >= -8 and (first + count).max() <= n_in + 8


# ------------------------------------------------------------- BiRefNet pre/post

def test_birefnet_prepare_image_kat():
    """/root/reference/test/test_segmentation.cpp:152-167 (BiRefNet.prepare_image): 4x3 RGBA image of 0..47."""
    from oracle import birefnet_oracle as B
    img = np.arange(4 * 3 * 4, dtype=np.uint8).reshape(3, 4, 4)
    t = B.prepare_image(img, mean=(0.4, 0.5, 0.6), std=(0.1, 0.2, 0.5))
    approx = lambda v: pytest.approx(v, rel=1e-5)       # Catch's Approx default
    assert t.shape == (1, 3, 3, 4) and t.dtype == np.float32
    assert t[0, 0, 0, 0] == approx(-4.0)
    assert t[0, 0, 0, 1] == approx((4.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 0, 0, 2] == approx((8.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 0, 1, 0] == approx((16.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 0, 1, 1] == approx((20.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 1, 1, 1] == approx((21.0 / 255.0 - 0.5) / 0.2)
    assert t[0, 2, 1, 1] == approx((22.0 / 255.0 - 0.6) / 0.5)


def test_birefnet_process_mask_kat():
    """/root/reference/test/test_segmentation.cpp:169-180 (BiRefNet.process_mask): expected values are the
    reference's own formula uint8_t(sigmoid(x) * 255) evaluated in float."""
    import math
    from oracle import birefnet_oracle as B
    values = np.array([0.0, 0.0, 0.2, -3.1, 0.0, 5.5, 0.0, 0.7, 0.0, 0.9], np.float32).reshape(2, 5)
    mask = B.process_mask(values)

    def expect(x):
        s = np.float32(1.0) / (np.float32(1.0) + np.float32(math.exp(-float(np.float32(x)))))
        return int(np.float32(s) * np.float32(255))
    assert mask.dtype == np.uint8 and mask.shape == (2, 5)
    assert mask[0, 0] == expect(0) == 127 and mask[0, 1] == expect(0)
    assert mask[0, 2] == expect(0.2) and mask[0, 3] == expect(-3.1)
    assert mask[1, 0] == expect(5.5) and mask[1, 2] == expect(0.7)


def test_resize_mask_box_filter_properties():
    """No upstream vector exists for resize_mask; these are the properties the published box (trapezoid) filter has:
    identity at scale 1, exact 2x2 means at 1/2, pixel replication at 2x, constant images stay constant."""
    from oracle import stb_resize as S
    rng = np.random.default_rng(0)
    m = rng.integers(0, 256, (16, 24), dtype=np.uint8)
    assert np.array_equal(S.resize_mask(m, 24, 16), m)
    half = S.resize_mask(m, 12, 8)
    mean = m.reshape(8, 2, 12, 2).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(half.astype(np.float64) - mean).max() <= 0.5 + 1e-6
    assert np.array_equal(S.resize_mask(m, 48, 32), np.repeat(np.repeat(m, 2, axis=0), 2, axis=1))
    assert np.all(S.resize_mask(np.full((7, 5), 200, np.uint8), 13, 3) == 200)
