"""Execution lanes and thread safety: 'Environment objects are safe to use from multiple threads'
(reference: src/include/dlimgedit/dlimgedit.hpp:98-101) must hold with several images in flight."""
import threading

import numpy as np
import pytest

from conftest import synthetic_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(model_dirs):
    from dlimgedit_amd import api
    mdir, params, cfg = model_dirs("vit_test")
    env = api.Environment(api.Options(api.Backend.gpu, mdir))
    return api, env


def test_lanes_exist_and_results_do_not_depend_on_the_lane(setup):
    api, env = setup
    assert api.ext.lane_count(env) >= 1
    img = synthetic_image(11)
    view = api.ImageView(img, api.Channels.rgba)
    # successive requests walk round-robin over the lanes: every lane must produce identical bits
    embs, masks = [], []
    for _ in range(2 * api.ext.lane_count(env) + 1):
        seg = api.Segmentation.process(view, env)
        embs.append(api.ext.get_embedding(seg))
        masks.append(seg.compute_mask(api.Point(400, 600)))
    for e, m in zip(embs[1:], masks[1:]):
        assert np.array_equal(e, embs[0]) and np.array_equal(m, masks[0])


def test_concurrent_threads_share_one_environment(setup):
    api, env = setup
    imgs = [synthetic_image(20 + i) for i in range(4)]
    want = []
    for im in imgs:
        seg = api.Segmentation.process(api.ImageView(im, api.Channels.rgba), env)
        want.append((api.ext.get_embedding(seg), seg.compute_mask(api.Point(512, 512))))
    results, errors = {}, []

    def worker(i):
        try:
            for _ in range(3):
                seg = api.Segmentation.process(api.ImageView(imgs[i], api.Channels.rgba), env)
                results[i] = (api.ext.get_embedding(seg), seg.compute_mask(api.Point(512, 512)))
        except Exception as e:       # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for i in range(4):
        assert np.array_equal(results[i][0], want[i][0]) and np.array_equal(results[i][1], want[i][1])


def test_async_steps_on_device_resident_images(setup):
    """The benchmark path: several encode_and_mask steps in flight over the lanes, masks equal the ABI path."""
    api, env = setup
    ext = api.ext
    imgs = [synthetic_image(30 + i) for i in range(3)]
    ptrs, mptrs = [], []
    for im in imgs:
        p = ext.device_alloc(env, im.nbytes)
        ext.copy_to_device(env, p, im)
        ptrs.append(p)
        mptrs.append(ext.device_alloc(env, 1024 * 1024))
    try:
        for _ in range(2):
            for p, m in zip(ptrs, mptrs):
                ext.encode_and_mask(env, ext.device_views([p], 1024, 1024), [api.Point(300, 300)], [m])
        ext.synchronize(env)
        for im, m in zip(imgs, mptrs):
            got = np.empty((1024, 1024), np.uint8)
            ext.copy_to_host(env, got, m)
            seg = api.Segmentation.process(api.ImageView(im, api.Channels.rgba), env)
            assert np.array_equal(got, seg.compute_mask(api.Point(300, 300)))
    finally:
        for p in ptrs + mptrs:
            ext.device_free(env, p)
