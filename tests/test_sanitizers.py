"""AddressSanitizer + UndefinedBehaviorSanitizer over the library's image file readers (csrc/image_io.cpp: PNG through
zlib; csrc/jpeg_decode.cpp: the JPEG decoder) on the CPU: the two pieces of the library that parse bytes it did not
produce (reference: stb_image behind /root/reference/src/image.cpp:11-23).  The sources are compiled as they lie in the
tree with ROCm's clang and -fsanitize=address,undefined (tests/sanitize/image_io_harness.cpp supplies main() and the
error helpers); well-formed files of every flavour the decoder takes must load, damaged ones (truncated anywhere, bytes
flipped, segment lengths overwritten) may be refused but must never trip a sanitizer."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CLANG = Path("/opt/rocm/lib/llvm/bin/clang++")
PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not CLANG.exists():
        pytest.skip("ROCm clang not found")
    out = tmp_path_factory.mktemp("sanitize") / "io_harness"
    csrc = ROOT / "dlimgedit_amd" / "csrc"
    cmd = [str(CLANG), "-std=c++17", "-O1", "-g", "-w", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}", f"-I{csrc}",
           str(ROOT / "tests" / "sanitize" / "image_io_harness.cpp"), str(csrc / "image_io.cpp"), str(csrc / "jpeg_decode.cpp"),
           "-lz", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitizer" in (r.stderr + r.stdout).lower() and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("this clang has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def run(harness, files):
    r = subprocess.run([str(harness), *map(str, files)], capture_output=True, text=True, errors="replace", timeout=600,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    loaded = [l for l in r.stdout.splitlines() if l.startswith("loaded ")]
    refused = [l for l in r.stdout.splitlines() if l.startswith("refused ")]
    assert len(loaded) + len(refused) == len(files)
    return loaded, refused


def sample_files(tmp):
    """Small well-formed files: JPEG baseline / progressive / restart intervals / grey / 4:2:0 / 4:4:4 / odd sizes, PNG
    grey / RGB / RGBA / palette / 16-bit / interlaced."""
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:67, 0:93]
    rgb = np.stack([(xx * 2.5) % 256, (yy * 3.1) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    rgb = np.clip(rgb.astype(np.int32) + rng.integers(-12, 12, rgb.shape), 0, 255).astype(np.uint8)
    files = []

    def jpeg(name, arr, **kw):
        p = tmp / name
        PIL.fromarray(arr).save(p, "JPEG", **kw)
        files.append(p)

    jpeg("base_420.jpg", rgb, quality=85, subsampling=2)
    jpeg("base_444.jpg", rgb, quality=92, subsampling=0)
    jpeg("base_422.jpg", rgb, quality=70, subsampling=1)
    jpeg("prog_420.jpg", rgb, quality=80, subsampling=2, progressive=True)
    jpeg("prog_444.jpg", rgb, quality=95, subsampling=0, progressive=True)
    jpeg("grey.jpg", rgb[:, :, 0], quality=88)
    jpeg("grey_prog.jpg", rgb[:, :, 1], quality=60, progressive=True)
    jpeg("restart.jpg", rgb, quality=85, subsampling=2, restart_marker_blocks=3)
    jpeg("tiny.jpg", rgb[:5, :7], quality=90)

    def png(name, img, **kw):
        p = tmp / name
        img.save(p, "PNG", **kw)
        files.append(p)

    png("rgb.png", PIL.fromarray(rgb))
    png("rgba.png", PIL.fromarray(np.dstack([rgb, 255 - rgb[:, :, 0]])))
    png("grey.png", PIL.fromarray(rgb[:, :, 2]))
    png("palette.png", PIL.fromarray(rgb).convert("P"))
    png("grey16.png", PIL.fromarray((rgb[:, :, 0].astype(np.uint16) * 257)))
    return files


def test_well_formed_files_load_clean(harness, tmp_path):
    files = sample_files(tmp_path) + [ROOT / "tests" / "golden" / "truck.jpg"]
    loaded, refused = run(harness, files)
    assert len(loaded) >= len(files) - 2, refused       # (16-bit / palette PNG flavours may be refused by design)
    for line in refused:
        assert "grey16.png" in line or "palette.png" in line, line


def test_damaged_files_never_trip_a_sanitizer(harness, tmp_path):
    rng = np.random.default_rng(99)
    originals = sample_files(tmp_path)
    truck = (ROOT / "tests" / "golden" / "truck.jpg").read_bytes()
    blobs = [(p.name, p.read_bytes()) for p in originals] + [("truck.jpg", truck[:60000])]
    damaged = []
    for name, data in blobs:
        n = len(data)
        stem, ext = name.rsplit(".", 1)
        variants = []
        for cut in sorted(set(int(c) for c in np.concatenate([np.arange(0, min(n, 40)), rng.integers(0, n, 25)]))):
            variants.append(data[:cut])                                     # truncated anywhere, header included
        for _ in range(40):                                                 # a few bytes flipped
            b = bytearray(data)
            for pos in rng.integers(0, n, int(rng.integers(1, 6))):
                b[pos] = int(rng.integers(0, 256))
            variants.append(bytes(b))
        for _ in range(20):                                                 # a run of bytes replaced by 0xFF / 0x00 / noise
            b = bytearray(data)
            pos = int(rng.integers(0, n))
            run_len = int(rng.integers(1, 64))
            fill = [b"\xff", b"\x00", None][int(rng.integers(0, 3))]
            b[pos:pos + run_len] = (fill * run_len) if fill else bytes(rng.integers(0, 256, run_len, dtype=np.uint8))
            variants.append(bytes(b[:n]))
        if ext == "jpg":                                                    # segment lengths and dimensions overwritten
            for marker in (b"\xff\xc0", b"\xff\xc2", b"\xff\xc4", b"\xff\xdb", b"\xff\xda", b"\xff\xdd"):
                at = data.find(marker)
                if at < 0:
                    continue
                for value in (0, 1, 2, 3, 0x7fff, 0xffff):
                    b = bytearray(data)
                    b[at + 2:at + 4] = value.to_bytes(2, "big")
                    variants.append(bytes(b))
                for off in (5, 7):                                          # height / width of a frame header
                    for value in (0, 1, 0xffff):
                        b = bytearray(data)
                        b[at + off:at + off + 2] = value.to_bytes(2, "big")
                        variants.append(bytes(b))
        else:                                                               # PNG: chunk lengths and IHDR fields, CRCs left stale
            for at in (8, 33):
                for value in (0, 1, 0x7fffffff, 0xffffffff):
                    b = bytearray(data)
                    b[at:at + 4] = value.to_bytes(4, "big")
                    variants.append(bytes(b))
            for field in (16, 20):                                          # width / height
                for value in (0, 1, 1 << 24, 0xffffffff):
                    b = bytearray(data)
                    b[field:field + 4] = value.to_bytes(4, "big")
                    variants.append(bytes(b))
        for i, v in enumerate(variants):
            p = tmp_path / f"{stem}_{i:03d}.{ext}"
            p.write_bytes(v)
            damaged.append(p)
    loaded, refused = run(harness, damaged)
    assert len(refused) > len(damaged) // 4          # most damage is noticed; what still loads is only required to be memory-safe


def _build(tmp, name, source, sanitize, extra=()):
    if not CLANG.exists():
        pytest.skip("ROCm clang not found")
    out = tmp / name
    cmd = [str(CLANG), "-std=c++17", "-O1", "-g", "-w", f"-fsanitize={sanitize}", "-fno-sanitize-recover=all",
           f"-I{ROOT / 'dlimgedit_amd' / 'csrc'}", str(ROOT / "tests" / "sanitize" / source), "-pthread", *extra, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in (r.stderr + r.stdout).lower():
        pytest.skip("this clang has no runtime for -fsanitize=" + sanitize)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def test_lane_worker_and_its_hand_over_protocols_under_thread_sanitizer(tmp_path):
    """csrc/lane_worker.hpp (the lanes' enqueue threads) under ThreadSanitizer: posts from several threads beside drains,
    the step queue's ticket protocol, the batch call's shared promise (tests/sanitize/lane_worker_tsan.cpp)."""
    exe = _build(tmp_path, "lane_worker_tsan", "lane_worker_tsan.cpp", "thread")
    r = subprocess.run([str(exe)], capture_output=True, text=True, errors="replace", timeout=600, env={"TSAN_OPTIONS": "halt_on_error=1"})
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_planners_on_random_inputs_under_address_sanitizer(tmp_path):
    """csrc/step_queue.hpp and csrc/mask_pieces.hpp on 220 000 random inputs and csrc/resize_tables.cpp on 1 500 axis pairs
    (1 pixel to 40 000, both filters) with ASan + UBSan, every result checked against the invariants its caller relies on
    (tests/sanitize/planners_fuzz.cpp)."""
    exe = _build(tmp_path, "planners_fuzz", "planners_fuzz.cpp", "address,undefined",
                 extra=["-ffp-contract=off", str(ROOT / "dlimgedit_amd" / "csrc" / "resize_tables.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, errors="replace", timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1000:], r.stderr[-4000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_damaged_weight_files_never_trip_a_sanitizer(tmp_path):
    """csrc/weights.cpp (the DLW parser: model files come from outside the library) under ASan + UBSan: an intact file of
    the reduced variant loads with every tensor readable end to end; files with the header, the geometry, the tensor
    table (dimensions, offsets, sizes -- including values whose sums or products wrap around 64 bits) or the tail damaged
    are refused or loaded, never a report."""
    import struct
    from dlimgedit_amd import weights as W
    from dlimgedit_amd.sam_config import get_config
    exe = _build(tmp_path, "weights_harness", "weights_harness.cpp", "address,undefined",
                 extra=["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                        str(ROOT / "dlimgedit_amd" / "csrc" / "weights.cpp")])
    cfg = get_config("vit_test")
    params = W.synthetic_weights(cfg, 3)
    full = tmp_path / "full.dlw"
    W.save_weights(full, cfg, params)
    whole = full.read_bytes()
    # a small file in the same format for the damaged variants: header and geometry of the real one, its first tensors of
    # modest size with their table entries re-pointed (entry: name[64], dtype, ndim, dims[4], offset, bytes)
    n_all = struct.unpack_from("<I", whole, 12)[0]
    picked = []
    for i in range(n_all):
        e = whole[80 + 120 * i: 80 + 120 * (i + 1)]
        off, nbytes = struct.unpack_from("<QQ", e, 104)
        if nbytes <= 32768 and len(picked) < 12:
            picked.append((e, whole[off:off + nbytes]))
    table, payload, at = b"", b"", 80 + 120 * len(picked)
    for e, blob in picked:
        table += e[:104] + struct.pack("<QQ", at + len(payload), len(blob))
        payload += blob
    data = whole[:12] + struct.pack("<I", len(picked)) + whole[16:80] + table + payload
    good = tmp_path / "good.dlw"
    good.write_bytes(data)
    count = len(picked)
    names = [data[80 + 120 * i: 80 + 120 * i + 64].split(b"\0")[0].decode() for i in range(count)]
    (tmp_path / "names.txt").write_text("\n".join(names) + "\n")
    rng = np.random.default_rng(17)
    files = [good]

    def add(blob):
        p = tmp_path / f"w_{len(files):04d}.dlw"
        p.write_bytes(blob)
        files.append(p)

    for cut in list(range(0, 90)) + [int(c) for c in rng.integers(90, len(data), 40)]:
        add(data[:cut])
    wild = [0, 1, 3, 4, 0x7fffffff, 0xffffffff, 1 << 32, (1 << 62), (1 << 63), (1 << 64) - 4, (1 << 64) - 1]
    for field in range(8, 80, 4):                           # version, count, geometry words
        for v in (0, 1, 0x7fffffff, 0xffffffff, 0x80000000):
            b = bytearray(data)
            struct.pack_into("<I", b, field, v)
            add(bytes(b))
    for entry in (0, 1, count // 2, count - 1):             # tensor table: dtype, ndim, dims, offset, size
        base = 80 + 120 * entry
        for off, fmt in ((64, "<I"), (68, "<I")):
            for v in (0, 1, 4, 5, 0xffffffff):
                b = bytearray(data)
                struct.pack_into(fmt, b, base + off, v)
                add(bytes(b))
        for off in (72, 80, 88, 96, 104, 112):
            for v in wild:
                b = bytearray(data)
                struct.pack_into("<Q", b, base + off, v)
                add(bytes(b))
        b = bytearray(data)                                 # dimensions whose product wraps to the true element count
        struct.pack_into("<QQ", b, base + 72, 1 << 63, 2)
        add(bytes(b))
    for _ in range(150):                                    # bytes flipped anywhere in header and table
        b = bytearray(data)
        for pos in rng.integers(0, 80 + 120 * count, int(rng.integers(1, 8))):
            b[pos] = int(rng.integers(0, 256))
        add(bytes(b))
    r = subprocess.run([str(exe), str(tmp_path / "names.txt"), *map(str, files)], capture_output=True, text=True, errors="replace", timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert lines[0].startswith(f"loaded {good}") and f": {count} tensors" in lines[0], lines[0]
    whole_run = subprocess.run([str(exe), str(tmp_path / "names.txt"), str(full)], capture_output=True, text=True, errors="replace", timeout=600)
    assert whole_run.returncode == 0 and whole_run.stdout.startswith(f"loaded {full}"), (whole_run.stdout[-500:], whole_run.stderr[-2000:])
    assert sum(l.startswith("refused ") for l in lines) > len(files) // 2
