"""Table slots 8 / 9 (load_image / save_image): PNG reader and writer of the library (csrc/image_io.cpp), host code,
no GPU needed.  Restates the reference's own "Image save" known-answer test (test/test_image.cpp:27-49: a 16x16 RGBA
image written, read back, every pixel compared) and pins the reader against files built here with zlib directly
(palette, sub-byte and 16-bit samples, Adam7) and the writer against an independent decode in Python."""
import struct
import zlib

import numpy as np
import pytest

from dlimgedit_amd import api


def _chunk(t, b):
    return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))


def _png(w, h, depth, ctype, rows, extra=b"", interlace=0):
    raw = b"".join(b"\0" + r for r in rows)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra +
            _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b""))


def _decode_reference(raw, ch):
    """Independent PNG decode (8-bit, non-interlaced) with zlib + the five row filters."""
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(raw):
        n = struct.unpack(">I", raw[pos:pos + 4])[0]
        t, body = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(t + body) == struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0]
        if t == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        if t == b"IDAT":
            idat += body
        pos += 12 + n
    dec = zlib.decompress(idat)
    stride = w * ch
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = dec[y * (stride + 1)]
        row = np.frombuffer(dec[y * (stride + 1) + 1:(y + 1) * (stride + 1)], np.uint8).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        for i in range(stride):
            a = cur[i - ch] if i >= ch else 0
            b = prev[i]
            c = prev[i - ch] if i >= ch else 0
            if ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            else:
                pred = (0, a, b, (a + b) // 2)[ft]
            cur[i] = (row[i] + pred) & 255
        out[y] = prev = cur
    return out.reshape(h, w, ch)


def test_reference_image_save_kat(tmp_path):
    """test/test_image.cpp:27-49."""
    px = np.zeros((16, 16, 4), np.uint8)
    for i in range(256):
        px.reshape(-1, 4)[i] = (255, i, 0, 255)
    path = tmp_path / "test_image_save.png"
    api.Image.save(api.ImageView(px, api.Channels.rgba), path)
    assert path.exists()
    result = api.Image.load(path)
    assert result.extent() == api.Extent(16, 16) and result.channels() == api.Channels.rgba
    assert np.array_equal(result.pixels(), px)


@pytest.mark.parametrize("ch,channels", [(4, api.Channels.rgba), (3, api.Channels.rgb), (1, api.Channels.mask)])
def test_round_trip_and_independent_decode(tmp_path, ch, channels):
    rng = np.random.default_rng(ch)
    a = rng.integers(0, 256, (37, 53, ch), dtype=np.uint8)
    a[5:20, 3:40] = (a[5:20, 3:40] // 32) * 32            # smooth-ish area: exercises the non-trivial row filters
    path = tmp_path / f"t{ch}.png"
    api.Image.save(api.ImageView(a if ch > 1 else a[:, :, 0], channels), path)
    assert np.array_equal(_decode_reference(path.read_bytes(), ch), a)
    img = api.Image.load(path)
    assert img.extent() == api.Extent(53, 37) and img.channels() == channels and img.size() == 37 * 53 * ch
    assert np.array_equal(img.pixels(), a)


def test_reader_on_foreign_files(tmp_path):
    rng = np.random.default_rng(0)
    pal = bytes([10, 20, 30, 40, 50, 60, 70, 80, 90, 100, 110, 120])
    idx = rng.integers(0, 4, (5, 7), dtype=np.uint8)
    rows = []
    for y in range(5):                                     # 2 bits per index, rows padded to a byte
        bits = "".join(format(int(v), "02b") for v in idx[y])
        bits += "0" * (-len(bits) % 8)
        rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    (tmp_path / "pal.png").write_bytes(_png(7, 5, 2, 3, rows, _chunk(b"PLTE", pal)))
    r = api.Image.load(tmp_path / "pal.png")
    assert r.channels() == api.Channels.rgb
    assert np.array_equal(r.pixels(), np.frombuffer(pal, np.uint8).reshape(4, 3)[idx])

    v16 = rng.integers(0, 65536, (4, 6, 3)).astype(">u2")
    (tmp_path / "rgb16.png").write_bytes(_png(6, 4, 16, 2, [v16[y].tobytes() for y in range(4)]))
    assert np.array_equal(api.Image.load(tmp_path / "rgb16.png").pixels(), (v16.astype(np.uint16) >> 8).astype(np.uint8))

    a = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)     # Adam7
    xs, ys, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    rows = []
    for p in range(7):
        sub = a[ys[p]::dy[p], xs[p]::dx[p]]
        rows += [sub[y].tobytes() for y in range(sub.shape[0])] if sub.size else []
    (tmp_path / "il.png").write_bytes(_png(11, 9, 8, 2, rows, interlace=1))
    assert np.array_equal(api.Image.load(tmp_path / "il.png").pixels(), a)

    g4 = rng.integers(0, 16, (3, 8), dtype=np.uint8)         # 4-bit grey scales to 0..255 in steps of 17
    (tmp_path / "g4.png").write_bytes(_png(8, 3, 4, 0, [bytes((int(g4[y, i]) << 4) | int(g4[y, i + 1]) for i in range(0, 8, 2))
                                                       for y in range(3)]))
    r = api.Image.load(tmp_path / "g4.png")
    assert r.channels() == api.Channels.mask and np.array_equal(r.pixels()[:, :, 0], g4 * 17)


def test_errors_follow_the_reference(tmp_path):
    rng = np.random.default_rng(1)
    with pytest.raises(api.Error, match="Failed to load image .*nope.png"):
        api.Image.load(tmp_path / "nope.png")
    ga = rng.integers(0, 256, (3, 3, 2), dtype=np.uint8)     # grey + alpha: 2 channels (reference: image.cpp:18-21)
    (tmp_path / "ga.png").write_bytes(_png(3, 3, 8, 4, [ga[y].tobytes() for y in range(3)]))
    with pytest.raises(api.Error, match=r"Unsupported number of channels \(2\)"):
        api.Image.load(tmp_path / "ga.png")
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)                 # a JPEG without a frame
    with pytest.raises(api.Error, match="Failed to load image .*x.jpg: "):
        api.Image.load(tmp_path / "x.jpg")
    bgra = api.ImageView(np.zeros((2, 2, 4), np.uint8), api.Channels.bgra)
    with pytest.raises(api.Error, match=r"Unsupported channel order \[5\]"):     # reference: image.cpp:26-29
        api.Image.save(bgra, tmp_path / "bgra.png")


def _photo_like(rng, w, h):
    """Smooth gradients + blobs + a little noise: what JPEG is made for (hard edges only measure the two libraries'
    different chroma filters)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        f = 110 + 60 * np.sin(xx * (0.02 + 0.01 * c) + c) * np.cos(yy * (0.015 + 0.004 * c)) + 40 * np.sin((xx + yy) * 0.007 * (c + 1))
        for _ in range(6):
            cx, cy, r, a = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(8, 60), rng.uniform(-70, 70)
            f += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
        img[:, :, c] = f + rng.normal(0, 2.0, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name,size,mode,save", [
    ("baseline 4:4:4", (64, 48), "RGB", dict(quality=90, subsampling=0)),
    ("baseline 4:2:0", (64, 48), "RGB", dict(quality=90, subsampling=2)),
    ("baseline 4:2:2", (64, 48), "RGB", dict(quality=85, subsampling=1)),
    ("odd size 4:2:0", (77, 53), "RGB", dict(quality=92, subsampling=2)),
    ("one pixel wide", (1, 37), "RGB", dict(quality=92, subsampling=2)),
    ("one pixel high", (45, 1), "RGB", dict(quality=92, subsampling=2)),
    ("grey", (50, 70), "L", dict(quality=88)),
    ("optimised tables", (96, 64), "RGB", dict(quality=75, subsampling=2, optimize=True)),
    ("progressive 4:2:0", (96, 80), "RGB", dict(quality=85, subsampling=2, progressive=True)),
    ("progressive 4:4:4", (70, 90), "RGB", dict(quality=95, subsampling=0, progressive=True)),
    ("progressive grey", (33, 47), "L", dict(quality=80, progressive=True)),
    ("restart markers", (120, 72), "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=3)),
    ("low quality", (64, 64), "RGB", dict(quality=20, subsampling=2)),
    ("full size", (1024, 683), "RGB", dict(quality=90, subsampling=2)),
])
def test_jpeg_files_load_like_a_jpeg_library_reads_them(tmp_path, name, size, mode, save):
    """load_image on JPEG files (the reference decodes them through stb_image, /root/reference/src/image.cpp:11-23): files
    written by Pillow in every flavour the decoder claims, compared with Pillow's own decode (libjpeg-turbo).  The two
    libraries round their inverse DCT and colour conversion differently and libjpeg-turbo's chroma filter differs at block
    edges, so the bar is a maximum difference of a few levels and a tight mean -- a decoder that mis-reads a Huffman code, a
    coefficient order, a sampling factor or a restart interval is off by tens of levels over whole blocks."""
    PILImage = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(zlib.crc32(name.encode()))      # (hash() of a str changes from process to process)
    w, h = size
    rgb = _photo_like(rng, w, h)
    src = PILImage.fromarray(rgb if mode == "RGB" else rgb[:, :, 1], mode)
    path = tmp_path / "t.jpg"
    try:
        src.save(path, "JPEG", **save)
    except TypeError:
        pytest.skip("this Pillow cannot write that flavour")
    want = np.asarray(PILImage.open(path).convert(mode))
    img = api.Image.load(path)
    got = img.pixels()
    assert img.channels() == (api.Channels.rgb if mode == "RGB" else api.Channels.mask)
    got = got if mode == "RGB" else got[:, :, 0]
    assert got.shape == want.shape
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 6 and diff.mean() < (0.8 if w * h >= 1000 else 1.5), (name, int(diff.max()), float(diff.mean()))
    # and it is the picture that went in, not merely something Pillow agrees with
    ref = rgb if mode == "RGB" else rgb[:, :, 1]
    assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).mean() < (12 if save.get("quality", 75) < 50 else 5)


def test_jpeg_flavours_that_are_refused_say_so(tmp_path):
    PILImage = pytest.importorskip("PIL.Image")
    cmyk = PILImage.fromarray(np.zeros((16, 16, 4), np.uint8), "CMYK")
    cmyk.save(tmp_path / "cmyk.jpg", "JPEG")
    with pytest.raises(api.Error, match="CMYK"):
        api.Image.load(tmp_path / "cmyk.jpg")
    good = tmp_path / "ok.jpg"
    PILImage.fromarray(np.full((40, 40, 3), 120, np.uint8), "RGB").save(good, "JPEG", quality=90)
    data = bytearray(good.read_bytes())
    big = tmp_path / "big.jpg"
    PILImage.fromarray(_photo_like(np.random.default_rng(3), 200, 200), "RGB").save(big, "JPEG", quality=90)
    whole = bytes(big.read_bytes())
    (tmp_path / "cut.jpg").write_bytes(whole[: len(whole) * 7 // 10])     # cut inside the scan: what is there is decoded
    cut, full = api.Image.load(tmp_path / "cut.jpg").pixels(), api.Image.load(big).pixels()
    assert cut.shape == full.shape and np.array_equal(cut[:64], full[:64])
    (tmp_path / "head.jpg").write_bytes(bytes(data[: len(data) // 2]))    # cut inside a table segment: an error, not a crash
    with pytest.raises(api.Error, match="Failed to load image"):
        api.Image.load(tmp_path / "head.jpg")
    huge = bytearray(data)
    i = huge.index(b"\xff\xc0")
    huge[i + 5:i + 9] = b"\x4e\x20\x4e\x20"                                # 20000 x 20000 in a 1 KB file
    (tmp_path / "huge.jpg").write_bytes(bytes(huge))
    with pytest.raises(api.Error, match="corrupt JPEG|too large"):
        api.Image.load(tmp_path / "huge.jpg")


def test_the_reference_s_own_jpeg_fixture():
    """tests/golden/truck.jpg is the one real input file of the reference's test suite (/root/reference/test/input/truck.jpg,
    1800 x 1200, baseline 4:2:0; the PNG fixtures there are git-LFS stubs).  The reference loads it with Image::load
    (test/test_segmentation.cpp).  Decoded here it matches Pillow's decode within 3 levels (mean 0.03), and its checksum pins
    the decoder's own arithmetic (integer IDCT, chroma filter, fixed-point colour conversion) against silent change."""
    from pathlib import Path
    path = Path(__file__).resolve().parent / "golden" / "truck.jpg"
    img = api.Image.load(path)
    got = img.pixels()
    assert got.shape == (1200, 1800, 3) and img.channels() == api.Channels.rgb
    assert zlib.crc32(got.tobytes()) == 0x57EDA8C9
    PILImage = pytest.importorskip("PIL.Image")
    want = np.asarray(PILImage.open(path).convert("RGB"))
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 3 and diff.mean() < 0.05
