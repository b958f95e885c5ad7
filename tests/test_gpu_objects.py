"""Pre- and post-processing of segment_objects (BiRefNet) on the device, through the C-ABI extension entry points,
bit-exact against the CPU oracle and against the reference's own known-answer tests
(/root/reference/test/test_segmentation.cpp:152-180).  SURVEY.md §8f rank 4."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from dlimgedit_amd import api
    return api.ext


@pytest.fixture(scope="module")
def api():
    from dlimgedit_amd import api
    return api


def test_prepare_image_reference_kat(ext, api):
    img = np.arange(4 * 3 * 4, dtype=np.uint8).reshape(3, 4, 4)
    t = ext.birefnet_prepare_image(img, api.Channels.rgba, (0.4, 0.5, 0.6), (0.1, 0.2, 0.5))
    approx = lambda v: pytest.approx(v, rel=1e-5)
    assert t[0, 0, 0, 0] == approx(-4.0)
    assert t[0, 0, 0, 1] == approx((4.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 0, 1, 0] == approx((16.0 / 255.0 - 0.4) / 0.1)
    assert t[0, 1, 1, 1] == approx((21.0 / 255.0 - 0.5) / 0.2)
    assert t[0, 2, 1, 1] == approx((22.0 / 255.0 - 0.6) / 0.5)


@pytest.mark.parametrize("w,h,channels", [(1024, 1024, 4), (333, 77, 3), (1, 1, 4), (2048, 1536, 3)])
def test_prepare_image_bit_exact(ext, api, w, h, channels):
    from oracle import birefnet_oracle as B
    rng = np.random.default_rng(w + h)
    img = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
    got = ext.birefnet_prepare_image(img, api.Channels.rgba if channels == 4 else api.Channels.rgb, B.BIREFNET_MEAN,
                                     B.BIREFNET_STD)
    assert np.array_equal(got, B.prepare_image(img))


def test_prepare_image_rejects_masks(ext, api):
    with pytest.raises(api.Error, match="three channels"):
        ext.birefnet_prepare_image(np.zeros((4, 4, 1), np.uint8), api.Channels.mask, (0, 0, 0), (1, 1, 1))


def test_process_mask_reference_kat(ext):
    values = np.array([0.0, 0.0, 0.2, -3.1, 0.0, 5.5, 0.0, 0.7, 0.0, 0.9], np.float32).reshape(2, 5)
    mask = ext.birefnet_process_mask(values)

    def expect(x):
        s = np.float32(1.0) / (np.float32(1.0) + np.float32(math.exp(-float(np.float32(x)))))
        return int(np.float32(s) * np.float32(255))
    assert mask[0, 0] == expect(0) and mask[0, 2] == expect(0.2) and mask[0, 3] == expect(-3.1)
    assert mask[1, 0] == expect(5.5) and mask[1, 2] == expect(0.7)


def test_process_mask_bit_exact(ext):
    from oracle import birefnet_oracle as B
    rng = np.random.default_rng(1)
    logits = (rng.standard_normal((1024, 1024)) * 6).astype(np.float32)
    logits[0, :8] = [0, -0.0, 88.0, -88.0, 104.0, -104.0, np.float32(1e-8), -30]      # saturation and tiny arguments
    assert np.array_equal(ext.birefnet_process_mask(logits), B.process_mask(logits))


@pytest.mark.parametrize("w,h,ow,oh", [(1024, 1024, 1800, 1200), (1024, 1024, 512, 341), (64, 48, 64, 48),
                                        (1024, 1024, 4000, 3000), (7, 5, 13, 3), (2048, 2048, 1024, 1024)])
def test_resize_mask_bit_exact(ext, w, h, ow, oh):
    from oracle import stb_resize as S
    rng = np.random.default_rng(w + oh)
    mask = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mask[: h // 2, : w // 2] = 255                    # a real mask is mostly flat
    assert np.array_equal(ext.resize_mask(mask, ow, oh), S.resize_mask(mask, ow, oh))
