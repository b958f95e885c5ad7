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
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    with pytest.raises(api.Error, match="JPEG decoding is not part"):
        api.Image.load(tmp_path / "x.jpg")
    bgra = api.ImageView(np.zeros((2, 2, 4), np.uint8), api.Channels.bgra)
    with pytest.raises(api.Error, match=r"Unsupported channel order \[5\]"):     # reference: image.cpp:26-29
        api.Image.save(bgra, tmp_path / "bgra.png")
