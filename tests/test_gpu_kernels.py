"""Kernel-level parity: every HIP kernel against the CPU oracle on seeded inputs, through the C-ABI
test hooks of include/dlimgedit/dlimgedit_amd.h.  Bit-exact for the byte/threshold kernels;
stated tolerances for f16-operand MFMA kernels (fp32 accumulate)."""
import numpy as np
import pytest

from conftest import synthetic_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    from dlimgedit_amd import api
    assert api.ext.device_count() >= 1, "no HIP device visible"
    assert api.Environment.is_supported(api.Backend.gpu), "GPU backend must be supported on the GPU box"
    return api.ext


@pytest.fixture(autouse=True)
def _product_tile_choice_after_each_test(ext):
    """Tests that force a GEMM tile (ext.force_gemm_tile) must not leak the choice into the next test."""
    yield
    ext.force_gemm_tile(-1)


def _oracle():
    from oracle import sam_oracle as O
    return O


# --------------------------------------------------------------------------------------------- K1

@pytest.mark.parametrize("channels,nbytes", [(4, 4), (5, 4), (6, 4), (3, 3), (1, 1)])
def test_preprocess_bit_exact(ext, channels, nbytes):
    """create_image_tensor KAT channel orders (reference test_segmentation.cpp:59-83) at full size:
    HIP output must equal f16(oracle fp32) bit for bit."""
    from dlimgedit_amd import api
    O = _oracle()
    img = synthetic_image(3, channels=4)
    if nbytes == 3:
        img = np.ascontiguousarray(img[:, :, :3])
    elif nbytes == 1:
        img = np.ascontiguousarray(img[:, :, :1])
    got = ext.test_preprocess(img, api.Channels(channels))
    want = O.patchify(O.preprocess(O.create_image_tensor(img, channels))).astype(np.float16)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("w,h", [(1024, 683), (683, 1024), (1024, 1), (37, 1024)])
def test_preprocess_padding_and_ragged(ext, w, h):
    """Non-square inputs: bottom/right region is zero in normalised space (graph-internal padding)."""
    from dlimgedit_amd import api
    O = _oracle()
    img = synthetic_image(5, width=w, height=h, channels=4)
    got = ext.test_preprocess(img, api.Channels.rgba)
    want = O.patchify(O.preprocess(O.create_image_tensor(img, 4))).astype(np.float16)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_preprocess_honours_stride(ext):
    """A 1024-wide view into a wider buffer: rows are taken `stride` bytes apart."""
    from dlimgedit_amd import api
    O = _oracle()
    full = synthetic_image(9, width=1100, height=1024, channels=4)
    out = np.empty((4096, 768), dtype=np.float16)
    rc = api.ext._h().dlimg_amd_test_preprocess(full.ctypes.data, 1024, 1024, 1100 * 4, 4, out.ctypes.data)
    assert rc == 0
    want = O.patchify(O.preprocess(O.create_image_tensor(np.ascontiguousarray(full[:, :1024]), 4))).astype(np.float16)
    assert np.array_equal(out.view(np.uint16), want.view(np.uint16))


# -------------------------------------------------------------------------------------------- K16

def _planes(seed, n=4):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    out = []
    for i in range(n):
        f = np.sin(xx * rng.uniform(0.02, 0.2) + rng.uniform(0, 6)) * np.cos(yy * rng.uniform(0.02, 0.2))
        out.append((f * 3 + rng.normal(0, 0.3, (256, 256))).astype(np.float32))
    return np.stack(out)


@pytest.mark.parametrize("w,h", [(1024, 1024), (1800, 1200), (512, 512), (640, 960), (1024, 768), (91, 37), (5, 3), (1024, 683),
                                 (681, 1024), (1024, 2)])
def test_postprocess_bit_exact(ext, w, h):
    """Two-stage bilinear + crop + threshold equals the oracle's masks bit for bit."""
    O = _oracle()
    planes = _planes(w * 7 + h, 1)
    got = ext.test_postprocess(planes, w, h)
    want = O.write_mask_image(O.postprocess_logits(planes[0], (h, w))[None, None], 0, (w, h))
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert set(np.unique(got)) <= {0, 255}


@pytest.mark.parametrize("w,h", [(1024, 1024), (1024, 683), (681, 1024), (1024, 2), (1022, 1024), (1024, 1021), (600, 1024)])
def test_postprocess_batched_identity_form_bit_exact(ext, w, h):
    """Launches of three or more masks whose second stage is the identity take the exact-4x form (16 x 8 pixels per lane,
    16-byte loads and stores; kernels/postprocess.hip): against the oracle, bit for bit, including the first two pixel
    rows / columns (clamped coordinate), the last ones (clamped index), extents that are not multiples of 16 or 8, and
    extreme logits at the borders of the plane."""
    O = _oracle()
    planes = _planes(w * 3 + h, 5)
    planes[1, 0, :] = 50.0
    planes[1, :, 0] = -50.0
    planes[2, 255, :] = np.linspace(-1e-3, 1e-3, 256, dtype=np.float32)
    planes[2, :, 255] = np.linspace(1e-3, -1e-3, 256, dtype=np.float32)
    planes[3] *= 1e-4
    got = ext.test_postprocess_batch(planes, w, h)
    for i in range(planes.shape[0]):
        want = O.write_mask_image(O.postprocess_logits(planes[i], (h, w))[None, None], 0, (w, h))
        assert np.array_equal(got[i], want), i
    assert set(np.unique(got)) <= {0, 255}


def test_postprocess_write_mask_kat(ext):
    """reference test_segmentation.cpp:85-99: strict > 0 (0.0 stays 0)."""
    planes = np.zeros((1, 256, 256), np.float32)
    planes[0, :, :128] = 0.2
    planes[0, :, 128:] = -3.1
    planes[0, 0, 0] = 0.0
    got = ext.test_postprocess(planes, 1024, 1024)
    O = _oracle()
    want = O.write_mask_image(O.postprocess_logits(planes[0], (1024, 1024))[None, None], 0, (1024, 1024))
    assert np.array_equal(got, want)
    assert got[500, 100] == 255 and got[500, 900] == 0


@pytest.mark.parametrize("iou,expect", [([0.9, 0.1, 0.5, 0.3], 2), ([0.2, 0.7, 0.7, 0.1], 1), ([600.0, 0.1, 0.2, 0.3], 0),
                                        ([0.0, -1.0, -2.0, -0.5], 3)])
def test_postprocess_single_mask_selection(ext, iou, expect):
    """SamOnnxModel.select_masks with two prompt tokens: token 0 is penalised by 500."""
    O = _oracle()
    planes = _planes(11, 4)
    iou = np.array(iou, np.float32)
    assert O.select_single(iou, 2) == expect
    got = ext.test_postprocess(planes, 1024, 1024, iou=iou)
    want = O.write_mask_image(O.postprocess_logits(planes[expect], (1024, 1024))[None, None], 0, (1024, 1024))
    assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------- GEMM

def _gemm_ref(A, W, bias, resid, act):
    O = _oracle()
    y = A.astype(np.float32) @ W.astype(np.float32).T
    if bias is not None:
        y = y + bias
    if act:
        y = O.gelu(y)
    if resid is not None:
        y = y + np.tile(resid, (A.shape[0] // resid.shape[0], 1))
    return y


@pytest.mark.parametrize("M,N,K,act,with_bias,resid_rows", [
    (128, 128, 64, 0, False, 0),        # single tile, single k-step
    (256, 192, 256, 0, True, 0),        # 64x64 tiles (N not a multiple of 128)
    (4096, 768, 768, 0, True, 4096),    # proj / patch-embed shape: 128x64 tiles, residual
    (4096, 2304, 768, 0, True, 0),      # qkv: 128x128 tiles
    (4096, 3072, 768, 1, True, 0),      # fc1 + GELU
    (4096, 768, 3072, 0, True, 4096),   # fc2
    (8192, 768, 768, 0, True, 4096),    # batch 2 with positional residual wrapping every 4096 rows
    (4096, 256, 2304, 0, False, 0),     # neck 3x3 as GEMM
    (16384, 128, 64, 1, True, 0),       # decoder upscaling stage 2
    (4096, 320, 320, 0, True, 0),       # head_dim-80 test width
])
def test_gemm_parity(ext, M, N, K, act, with_bias, resid_rows):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32) if with_bias else None
    resid = rng.standard_normal((resid_rows, N)).astype(np.float32) if resid_rows else None
    out32, out16 = ext.test_gemm(A, W, bias, resid, act, want_f16=True)
    ref = _gemm_ref(A, W, bias, resid, act)
    # fp32 accumulation of exact f16 products: only summation order differs from the fp32 BLAS
    assert np.abs(out32 - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    assert np.array_equal(out16, out32.astype(np.float16))


@pytest.mark.parametrize("tile,M,N,K", [(0, 256, 768, 192), (1, 256, 576, 256), (2, 256, 256, 128), (3, 256, 192, 320),
                                        (4, 256, 128, 64), (5, 128, 64, 128), (6, 512, 512, 192), (6, 256, 256, 64),
                                        (7, 512, 512, 192), (7, 256, 256, 64), (8, 256, 384, 320),
                                        (9, 512, 512, 192), (9, 256, 256, 64), (9, 256, 512, 128), (9, 512, 256, 448),
                                        (9, 768, 512, 1024),
                                        (10, 256, 512, 192), (10, 128, 256, 64), (10, 384, 256, 448), (10, 256, 768, 1024),
                                        (10, 128, 256, 128),
                                        (11, 192, 512, 192), (11, 64, 256, 64), (11, 320, 256, 448), (11, 128, 768, 1024),
                                        (11, 64, 256, 128)])
def test_gemm_every_tile_configuration(ext, monkeypatch, tile, M, N, K):
    """Each tile configuration (waves layout, K-tile, pipeline depth) against the fp32 product, incl. K tails
    shorter than the pipeline depth."""
    ext.force_gemm_tile(tile)
    rng = np.random.default_rng(tile * 1000 + K)
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    out = ext.test_gemm(A, W, bias, resid, 0)
    ref = _gemm_ref(A, W, bias, resid, 0)
    assert np.abs(out - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())


def test_gemm_identity_layout(ext):
    """A = I against an asymmetric W catches a transposed or permuted C write."""
    K = 128
    A = np.zeros((128, K), np.float16)
    A[np.arange(128), np.arange(128)] = 1
    W = (np.arange(192 * K).reshape(192, K) % 251).astype(np.float16)
    out = ext.test_gemm(A, W)
    assert np.array_equal(out, W.astype(np.float32).T[:128])


@pytest.mark.parametrize("M,D,K1,N,act,tile", [
    (4096, 768, 768, 2304, 0, None),       # ViT-B proj -> LN1 -> qkv
    (4096, 768, 3072, 3072, 1, None),      # ViT-B fc2 -> LN2 -> fc1 + GELU
    (4096, 1280, 1280, 3840, 0, None),     # ViT-H width (40 statistic groups per row)
    (4096, 128, 768, 384, 0, None),        # test width, patch embedding as producer
    (4096, 320, 320, 1280, 1, None),       # head_dim-80 test width
    (512, 1024, 256, 512, 0, 6),           # 32x32x16 256x256 tile on both sides
    (512, 1024, 256, 512, 1, 7),           # 16x16x32 256x256 tile on both sides
    (512, 1024, 256, 512, 1, 9),           # ping-pong 256x256 kernel on both sides
    (4096, 768, 768, 2304, 0, 9),          # ... at ViT-B's proj -> LN1 -> qkv
    (512, 1024, 256, 512, 1, 10),          # 128x256 ping-pong kernel on both sides
    (4096, 768, 3072, 768, 0, 10),         # ... as ViT-B's fc2 (producer of the stream and its statistics)
    (512, 1024, 256, 512, 1, 11),          # 64x256 ping-pong kernel on both sides
    (4096, 768, 3072, 768, 0, 11),         # ... as ViT-B's fc2 when a pass has the GPU to itself
    (256, 256, 128, 256, 0, 2),
    (256, 256, 128, 384, 0, 8),
])
def test_gemm_folded_layernorm(ext, monkeypatch, M, D, K1, N, act, tile):
    """LayerNorm folded into the GEMMs around it: the producer leaves the f16 copy of the stream, the consumer
    multiplies it by W*gamma, takes the row moments from its own operand fragments and normalises in its epilogue.
    Checked against the oracle's LayerNorm + fp64 product and against the un-fused kernels' own error."""
    if tile is not None:
        ext.force_gemm_tile(tile)
    O = _oracle()
    rng = np.random.default_rng(M + D + N)
    A1 = rng.standard_normal((M, K1)).astype(np.float16)
    W1 = (rng.standard_normal((D, K1)) / np.sqrt(K1)).astype(np.float16)
    b1 = rng.standard_normal(D).astype(np.float32)
    resid = (rng.standard_normal((M, D)) * 3 + 1).astype(np.float32)
    resid[:, 5] += 60.0                      # one massive-activation channel, as trained ViTs have them
    gamma = (1 + 0.2 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(D)).astype(np.float32)
    W2 = (rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
    b2 = rng.standard_normal(N).astype(np.float32)

    x, xh, y = ext.test_gemm_ln(A1, W1, b1, resid, W2, gamma, beta, b2, 1e-6, act)
    x_ref = _gemm_ref(A1, W1, b1, resid, 0)
    assert np.abs(x - x_ref).max() <= 2e-3 * max(1.0, np.abs(x_ref).max())
    assert np.array_equal(xh, x.astype(np.float16))

    xn = O.layer_norm(x.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64), 1e-6)
    ref = xn @ W2.astype(np.float64).T + b2
    if act:
        ref = O.gelu(ref)
    err = y - ref
    # the same two steps as separate kernels (LayerNorm -> f16, GEMM with f16 W) bound what f16 operands cost
    _, xn16 = ext.test_layernorm(x, gamma, beta, 1e-6)
    err_split = ext.test_gemm(xn16, W2.astype(np.float16), b2, None, act) - ref
    rms, rms_split = np.sqrt((err ** 2).mean()), np.sqrt((err_split ** 2).mean())
    assert rms <= 2.0 * rms_split + 1e-4, (rms, rms_split)
    assert np.abs(err).max() <= 2e-2, np.abs(err).max()

    _, _, y2 = ext.test_gemm_ln(A1, W1, b1, resid, W2, gamma, beta, b2, 1e-6, act)
    assert np.array_equal(y, y2)             # fixed summation order, no atomics


def test_gemm_folded_layernorm_offset_rows(ext):
    """Rows whose mean is large against their spread (|mean| = 30 sigma): the raw-moment variance still holds."""
    O = _oracle()
    rng = np.random.default_rng(5)
    M, D, N = 256, 768, 256
    x0 = (rng.standard_normal((M, D)) + 30.0).astype(np.float32)
    eye = np.eye(D, dtype=np.float16)
    gamma, beta = np.ones(D, np.float32), np.zeros(D, np.float32)
    W2 = (rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
    x, xh, y = ext.test_gemm_ln(np.zeros((M, D), np.float16), eye, None, x0, W2, gamma, beta, np.zeros(N, np.float32),
                                1e-6, 0)
    assert np.array_equal(x, x0)
    # the consumer normalises the f16 stream it multiplies with: that is the reference
    ref = O.layer_norm(xh.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64), 1e-6) \
        @ W2.astype(np.float16).astype(np.float64).T
    assert np.abs(y - ref).max() <= 5e-3, np.abs(y - ref).max()


def test_gemm_rejects_bad_shapes(ext):
    from dlimgedit_amd import api
    A = np.zeros((100, 64), np.float16)
    W = np.zeros((64, 64), np.float16)
    with pytest.raises(api.Error, match="multiples of 64"):
        ext.test_gemm(A, W)


# -------------------------------------------------------------------------------------- LayerNorm

@pytest.mark.parametrize("rows,dim,act", [(4096, 768, 0), (4096, 1280, 0), (4096, 256, 0), (16384, 64, 1), (7, 256, 0),
                                          (4096, 128, 0), (4096, 320, 0)])
def test_layernorm_parity(ext, rows, dim, act):
    O = _oracle()
    rng = np.random.default_rng(rows + dim)
    x = (rng.standard_normal((rows, dim)) * 3 + 1).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float32)
    b = (0.1 * rng.standard_normal(dim)).astype(np.float32)
    o32, o16 = ext.test_layernorm(x, w, b, 1e-6, act)
    ref = O.layer_norm(x, w, b, 1e-6)
    if act:
        ref = O.gelu(ref)
    assert np.abs(o32 - ref).max() < 2e-5
    assert np.array_equal(o16, o32.astype(np.float16))


# -------------------------------------------------------------------------------------- attention

def _qkv(seed, heads, hd, batch=1):
    rng = np.random.default_rng(seed)
    D = heads * hd
    qkv = rng.standard_normal((batch * 4096, 3 * D)).astype(np.float16)
    bias = (0.5 * rng.standard_normal(3 * D)).astype(np.float32)
    return qkv, bias


@pytest.mark.parametrize("heads,hd", [(2, 64), (2, 80)])
def test_attention_window_parity(ext, heads, hd):
    """Windowed attention incl. zero-padded tokens as un-masked keys and decomposed rel-pos bias."""
    O = _oracle()
    qkv, bias = _qkv(1, heads, hd)
    rng = np.random.default_rng(2)
    rel_h = (0.2 * rng.standard_normal((27, hd))).astype(np.float32)
    rel_w = (0.2 * rng.standard_normal((27, hd))).astype(np.float32)
    got = ext.test_attention(False, qkv, bias, rel_h, rel_w, 1, heads, hd).astype(np.float32)
    # the kernel sees the bias rounded to f16 for pad tokens, as the GEMM epilogue would have stored it
    ref = O.windowed_attention_from_qkv(qkv.astype(np.float32), bias.astype(np.float16).astype(np.float32),
                                        rel_h.astype(np.float16).astype(np.float32),
                                        rel_w.astype(np.float16).astype(np.float32), heads)
    err = np.abs(got - ref).max()
    assert err < 6e-3, err          # P and V are f16 operands of the second MFMA; output rounded to f16


@pytest.mark.gpu
@pytest.mark.parametrize("hd", [64, 80])
def test_attention_window_lazy_maximum_rescale(ext, hd):
    """Both head dimensions make one pass with a reference maximum taken from the first key tile: keys late in every window
    (last window row) aligned with a query and far above the rest force the rescale branch (accumulators and row sum, 80
    columns at head dimension 80), in windows with and without zero padding; the softmax must come out as the oracle's
    exact-maximum one."""
    O = _oracle()
    heads = 2
    qkv, bias = _qkv(11, heads, hd)
    qkv = (qkv.astype(np.float32) * 0.3).astype(np.float16)
    D = heads * hd
    for wy in range(5):
        for wx in range(5):
            ty, tx = (13, 5) if wy < 4 else (7, 5)          # a real token late in the window's key order
            gy, gx = wy * 14 + ty, wx * 14 + tx
            if gx >= 64:
                gx = wx * 14 + 3
            q_tok = (wy * 14) * 64 + min(wx * 14 + 2, 63)
            k_tok = gy * 64 + gx
            for h in range(heads):
                qv = qkv[q_tok, h * hd:(h + 1) * hd].astype(np.float32)
                qkv[k_tok, D + h * hd:D + (h + 1) * hd] = (qv * 60).astype(np.float16)
    rng = np.random.default_rng(12)
    rel_h = (0.2 * rng.standard_normal((27, hd))).astype(np.float32)
    rel_w = (0.2 * rng.standard_normal((27, hd))).astype(np.float32)
    got = ext.test_attention(False, qkv, bias, rel_h, rel_w, 1, heads, hd).astype(np.float32)
    ref = O.windowed_attention_from_qkv(qkv.astype(np.float32), bias.astype(np.float16).astype(np.float32),
                                        rel_h.astype(np.float16).astype(np.float32),
                                        rel_w.astype(np.float16).astype(np.float32), heads)
    # the forcing keys really are beyond the lazy threshold (2^8 in the exponent): |q|^2 * 60 / sqrt(hd) * log2(e) >> 8
    q0 = qkv[0 * 64 + 2, :hd].astype(np.float32)
    assert float(q0 @ q0) * 60 / np.sqrt(hd) * 1.4427 > 16
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 6e-3, np.abs(got - ref).max()


@pytest.mark.parametrize("heads,hd,batch", [(2, 64, 1), (2, 80, 1), (1, 64, 2)])
def test_attention_global_parity(ext, heads, hd, batch):
    """Flash-style global attention with on-the-fly rel-pos bias against the explicit 4096x4096 softmax."""
    O = _oracle()
    qkv, _ = _qkv(3, heads, hd, batch)
    rng = np.random.default_rng(4)
    rel_h = (0.2 * rng.standard_normal((127, hd))).astype(np.float32)
    rel_w = (0.2 * rng.standard_normal((127, hd))).astype(np.float32)
    got = ext.test_attention(True, qkv, None, rel_h, rel_w, batch, heads, hd).astype(np.float32)
    ref = O.attention_from_qkv(qkv.astype(np.float32).reshape(batch, 4096, -1),
                               rel_h.astype(np.float16).astype(np.float32),
                               rel_w.astype(np.float16).astype(np.float32), heads, 64).reshape(batch * 4096, -1)
    err = np.abs(got - ref).max()
    assert err < 6e-3, err


def test_attention_global_online_softmax_rescale(ext):
    """Force the running-max rescale branch: one key row far above the rest, late in the stream."""
    O = _oracle()
    heads, hd = 1, 64
    qkv, _ = _qkv(8, heads, hd)
    qkv = (qkv.astype(np.float32) * 0.3).astype(np.float16)
    qkv[3000, hd:2 * hd] = (qkv[5, 0:hd].astype(np.float32) * 40).astype(np.float16)   # k_3000 aligned with q_5
    rel = np.zeros((127, hd), np.float32)
    got = ext.test_attention(True, qkv, None, rel, rel, 1, heads, hd).astype(np.float32)
    ref = O.attention_from_qkv(qkv.astype(np.float32)[None], rel, rel, heads, 64)[0]
    assert np.abs(got - ref).max() < 6e-3


# -------------------------------------------------------------------------------------------- K17

@pytest.mark.parametrize("w,h,ow,oh,channels", [
    (8, 8, 4, 4, 4),            # the reference's KAT geometry
    (1800, 1200, 1024, 683, 3), # truck.jpg geometry: downsample, rgb
    (512, 512, 1024, 1024, 4),  # cat_and_hat geometry: 2x upsample (Catmull-Rom)
    (640, 960, 683, 1024, 4),   # mild upsample, portrait
    (2048, 1536, 1024, 768, 1), # 2x downsample, mask
    (37, 91, 416, 1024, 5),     # large upsample of a tiny bgra image
])
def test_resize_bit_exact(ext, w, h, ow, oh, channels):
    """The device resampler equals the oracle's restatement of stbir_resize_uint8_generic byte for byte."""
    from dlimgedit_amd import api
    from oracle import stb_resize as R
    c = api.count(api.Channels(channels))
    rng = np.random.default_rng(w * 31 + h)
    img = synthetic_image(w + h, width=w, height=h, channels=4)[:, :, :c].copy()
    img[rng.integers(0, h, 50), rng.integers(0, w, 50)] = rng.integers(0, 256, (50, c))   # hard edges
    got = ext.test_resize(img, api.Channels(channels), ow, oh)
    want = R.resize_srgb(img, ow, oh)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_resize_upsampling_against_pillow_on_device(ext):
    """The device resampler against the fixture that does not come from oracle/stb_resize.py (Pillow Catmull-Rom in
    linear light, tests/golden/make_resize_golden.py): within 1 LSB everywhere."""
    from dlimgedit_amd import api
    from test_oracle_kats import _check_against_pillow_upsample
    _check_against_pillow_upsample(
        lambda src, ow, oh: ext.test_resize(src, api.Channels.rgb if src.shape[2] == 3 else api.Channels.rgba, ow, oh))


def test_resize_reference_kat_on_device(ext):
    from dlimgedit_amd import api
    img = np.zeros((8, 8, 4), np.uint8)
    for i in range(64):
        img[i // 8, i % 8] = (255, 4 * (i // 8), 4 * (i % 8), 255)
    out = ext.test_resize(img, api.Channels.rgba, 4, 4)
    for i in range(16):
        assert tuple(int(v) for v in out[i // 4, i % 4]) == (255, 2 + 8 * (i // 4), 2 + 8 * (i % 4), 255)


@pytest.mark.parametrize("N,K,act", [(2304, 128, 0), (3072, 256, 1), (1536, 1152, 0)])
def test_gemm_many_rounds_of_tiles_bit_equal_to_single_rounds(ext, N, K, act):
    """Launches with more tiles than CUs (batched passes: two images give qkv 288 and fc1 384 tiles of 256 x 256, eight images
    1152 / 1536) against the same rows in launches of one round: which workgroup (or, in the persistent form of r04 that
    these tests were written for, which turn of a workgroup's tile loop) computes a tile must not enter the arithmetic.
    M = 16384 rows in one launch (576 / 768 / 384 tiles) against four launches of 4096 rows, bit for bit, for the plain,
    the GELU and the residual-stream flavours; and against the fp32 product."""
    ext.force_gemm_tile(9)
    rng = np.random.default_rng(N + K)
    M = 16384
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32) if N == 1536 else None
    whole32, whole16 = ext.test_gemm(A, W, bias, resid, act, want_f16=True)
    ref = _gemm_ref(A[:512], W, bias, None if resid is None else resid[:512], act)
    assert np.abs(whole32[:512] - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    for q in range(4):
        rows = slice(q * 4096, (q + 1) * 4096)
        part32, part16 = ext.test_gemm(A[rows], W, bias, None if resid is None else resid[rows], act, want_f16=True)
        assert np.array_equal(part32, whole32[rows]) and np.array_equal(part16, whole16[rows]), q


def test_gemm_many_rounds_of_tiles_with_folded_layernorm(ext):
    """The same for the LayerNorm-folded consumer: 8192 rows (producer 96 tiles, consumer 288) against the two halves run on
    their own -- row statistics are merged per tile from the producer's partials, so a half must see exactly its rows'."""
    ext.force_gemm_tile(9)
    rng = np.random.default_rng(99)
    M, D, K1, N = 8192, 768, 256, 2304
    A1 = rng.standard_normal((M, K1)).astype(np.float16)
    W1 = (rng.standard_normal((D, K1)) / np.sqrt(K1)).astype(np.float16)
    b1 = rng.standard_normal(D).astype(np.float32)
    resid = (rng.standard_normal((M, D)) * 3 + 1).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(D)).astype(np.float32)
    W2 = (rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
    b2 = rng.standard_normal(N).astype(np.float32)
    for act in (0, 1):
        x, xh, y = ext.test_gemm_ln(A1, W1, b1, resid, W2, gamma, beta, b2, 1e-6, act)
        for h in range(2):
            rows = slice(h * 4096, (h + 1) * 4096)
            xp, xhp, yp = ext.test_gemm_ln(A1[rows], W1, b1, resid[rows], W2, gamma, beta, b2, 1e-6, act)
            assert np.array_equal(xp, x[rows]) and np.array_equal(xhp, xh[rows]) and np.array_equal(yp, y[rows]), (act, h)


def test_ping_pong_tiles_compute_the_same_bits(ext):
    """Tiles 9 (256 x 256), 10 (128 x 256) and 11 (64 x 256) of kernels/gemm.hip use the same MFMA, K order, epilogue
    arithmetic and 64-column statistics groups, so WHICH of them writes the residual stream (two images: 9; one image
    beside other lanes: 10; one image with the GPU to itself: 11) must not change a bit of the stream, of its f16 copy or
    of the row statistics it leaves -- the latter seen through the LayerNorm-folded consumer, which the product always
    runs on tile 9 for these shapes (gemm_pick_tile keeps a consumer's tile independent of the pass: its merge of the
    statistics depends on the tile height).  ViT-B's fc2 -> LN -> qkv shapes, with and without GELU."""
    rng = np.random.default_rng(2024)
    M, D, K1, N = 1024, 768, 3072, 2304
    A1 = rng.standard_normal((M, K1)).astype(np.float16)
    W1 = (rng.standard_normal((D, K1)) / np.sqrt(K1)).astype(np.float16)
    b1 = rng.standard_normal(D).astype(np.float32)
    resid = (rng.standard_normal((M, D)) * 3 + 1).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(D)).astype(np.float32)
    W2 = (rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
    b2 = rng.standard_normal(N).astype(np.float32)
    for act in (0, 1):
        results = []
        for tile in (9, 10, 11):
            ext.force_gemm_tile(tile, consumer_tile=9)
            results.append(ext.test_gemm_ln(A1, W1, b1, resid, W2, gamma, beta, b2, 1e-6, act))
        for other, tile in zip(results[1:], (10, 11)):
            for a, b, what in zip(results[0], other, ("stream", "f16 copy", "consumer (statistics)")):
                assert np.array_equal(a, b), (act, tile, what)


@pytest.mark.parametrize("tile", [9, 10, 11])
@pytest.mark.parametrize("with_resid", [False, True])
def test_pair_stream_is_the_fp32_stream_split_in_two(ext, tile, with_resid):
    """The residual stream as an f16 pair (GemmArgs::out_l / resid_h / resid_l, kernels/gemm.hip): hi = f16(x) -- the f16 copy
    the fp32 representation leaves as well -- and lo = f16(x - hi), where x is the fp32 value the other representation
    stores.  Same residual (hi + lo, summed in fp32 either way), so: hi and the row statistics equal bit for bit, lo equal to
    the split of the fp32 result done here, and hi + lo within 2^-22 relative of x.  ViT-B's fc2 shape, every ping-pong tile."""
    rng = np.random.default_rng(77 + tile)
    M, D, K = 1024, 768, 3072
    A = rng.standard_normal((M, K)).astype(np.float16)
    W = (rng.standard_normal((D, K)) / np.sqrt(K)).astype(np.float16)
    bias = rng.standard_normal(D).astype(np.float32)
    rh = rl = None
    if with_resid:
        r = (rng.standard_normal((M, D)) * 3 + 1).astype(np.float32)
        r[::7, ::5] *= 40                                       # the few large channels a SAM stream carries
        rh = r.astype(np.float16)
        rl = (r - rh.astype(np.float32)).astype(np.float16)
    ext.force_gemm_tile(tile)
    x, hi32, _, st32 = ext.test_gemm_stream(A, W, bias, rh, rl, pair=False)
    _, hi, lo, st = ext.test_gemm_stream(A, W, bias, rh, rl, pair=True)
    assert np.array_equal(hi.view(np.uint16), hi32.view(np.uint16))
    assert np.array_equal(st.view(np.uint32), st32.view(np.uint32))
    assert np.array_equal(hi, x.astype(np.float16))
    want_lo = (x - hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(lo.view(np.uint16), want_lo.view(np.uint16))
    back = hi.astype(np.float32) + lo.astype(np.float32)
    assert np.all(np.abs(back - x) <= np.abs(x) * 2.0 ** -21 + 2.0 ** -24)
