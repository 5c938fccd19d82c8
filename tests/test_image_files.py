"""Image files behind the host library: `image::open(path)?.to_rgb8()` of Image::load (src/textures/image.rs:17-26)
and RgbImage::save("*.png") (src/main.rs:93).  The `image` crate is not part of the reference tree, so the files are
checked against an independent statement of the PNG format in this file (Python's zlib and struct: chunks, CRCs,
the five scan-line filters), in both directions, and against the PNG specification's own known values."""
import struct
import zlib

import numpy as np
import pytest

from pathtracer_amd.core import PtError, load_image, save_png
from pathtracer_amd.scene_src import SceneSource


def _chunk(kind: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _write_png(path, rows: np.ndarray, color: int, depth: int, filters, palette=None, split_idat=1, extra=b""):
    """rows: (h, stride) uint8 of packed scan lines; every row is filtered with filters[y % len(filters)]."""
    h, stride = rows.shape
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bpp = max(1, channels * depth // 8)
    width = stride * 8 // (channels * depth)
    raw = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        f = filters[y % len(filters)]
        out = np.zeros(stride, np.int32)
        for x in range(stride):
            a = cur[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            pred = [0, a, b, (a + b) // 2, _paeth(int(a), int(b), int(c))][f]
            out[x] = (cur[x] - pred) & 255
        raw.append(f)
        raw += bytes(out.astype(np.uint8))
        prev = cur
    z = zlib.compress(bytes(raw), 9)
    parts = [z[i * len(z) // split_idat:(i + 1) * len(z) // split_idat] for i in range(split_idat)]
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, h, depth, color, 0, 0, 0)) + extra
    if palette is not None:
        data += _chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    for p in parts:
        data += _chunk(b"IDAT", p)
    data += _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)
    return data


def _read_png_rgb8(path):
    """8-bit RGB, filter None on every row: what save_png writes."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, head = 8, b"", None
    kinds = []
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body) & 0xFFFFFFFF
        kinds.append(kind)
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        if kind == b"IDAT":
            idat += body
        at += 12 + n
    assert kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
    w, h, depth, color, comp, filt, lace = head
    assert (depth, color, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def test_rgb_png_with_every_filter_and_split_idat(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    img[3:9, 2:11] = (np.arange(9)[None, :, None] * 20 + np.arange(6)[:, None, None] * 7).astype(np.uint8)  # smooth part: filters matter
    p = str(tmp_path / "a.png")
    _write_png(p, img.reshape(13, -1), 2, 8, [0, 1, 2, 3, 4], split_idat=3, extra=_chunk(b"tEXt", b"Comment\x00hello"))
    assert np.array_equal(load_image(p), img)


def test_grey_alpha_palette_and_16_bit_become_rgb8(tmp_path):
    rng = np.random.default_rng(2)
    h, w = 9, 12
    grey = rng.integers(0, 256, (h, w), dtype=np.uint8)
    p = str(tmp_path / "g.png")
    _write_png(p, grey, 0, 8, [4, 3])
    assert np.array_equal(load_image(p), np.repeat(grey[..., None], 3, axis=2))
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    _write_png(p, rgba.reshape(h, -1), 6, 8, [1, 4, 2])
    assert np.array_equal(load_image(p), rgba[..., :3])  # to_rgb8 drops alpha, it does not blend
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    _write_png(p, ga.reshape(h, -1), 4, 8, [3])
    assert np.array_equal(load_image(p), np.repeat(ga[..., :1], 3, axis=2))
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    _write_png(p, idx, 3, 8, [0, 2], palette=pal)
    assert np.array_equal(load_image(p), pal[idx])
    packed = (idx[:, 0::2] << 4 | idx[:, 1::2]).astype(np.uint8)  # 4 bits per index, leftmost pixel in the high bits
    _write_png(p, packed, 3, 4, [0, 1], palette=pal)
    assert np.array_equal(load_image(p), pal[idx])
    bits = rng.integers(0, 2, (h, 16), dtype=np.uint8)
    _write_png(p, np.packbits(bits, axis=1), 0, 1, [0])
    assert np.array_equal(load_image(p), np.repeat((bits * 255)[..., None], 3, axis=2))
    deep = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    deep[0, 0] = (0, 65535, 32768)
    _write_png(p, deep.astype(">u2").view(np.uint8).reshape(h, -1), 2, 16, [4, 1])
    want = ((deep.astype(np.uint32) + 128) // 257).astype(np.uint8)
    got = load_image(p)
    assert np.array_equal(got, want) and tuple(got[0, 0]) == (0, 255, 128)


def test_png_written_by_the_library_reads_back_both_ways(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (21, 34, 3), dtype=np.uint8)
    p = str(tmp_path / "out.png")
    save_png(p, img)
    assert np.array_equal(_read_png_rgb8(p), img)
    assert np.array_equal(load_image(p), img)


def test_ppm_binary_plain_and_comments(tmp_path):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    p = str(tmp_path / "a.ppm")
    open(p, "wb").write(b"P6\n# a comment\n7 5\n255\n" + img.tobytes())
    assert np.array_equal(load_image(p), img)
    open(p, "wb").write(b"P3 7 5 # size\n255\n" + " ".join(str(v) for v in img.reshape(-1)).encode() + b"\n")
    assert np.array_equal(load_image(p), img)
    open(p, "wb").write(b"P6 2 1 15\n" + bytes([0, 15, 8, 3, 3, 3]))
    assert load_image(p).reshape(-1).tolist() == [0, 255, 136, 51, 51, 51]


def test_failures_are_reported_and_fall_back_to_the_default_texture(tmp_path, capfd):
    with pytest.raises(PtError):
        load_image(str(tmp_path / "missing.png"))
    p = str(tmp_path / "bad.png")
    rows = np.zeros((4, 12), np.uint8)
    data = bytearray(_write_png(p, rows, 2, 8, [0]))
    data[40] ^= 0xFF  # inside the IDAT chunk: the CRC no longer matches
    open(p, "wb").write(bytes(data))
    with pytest.raises(PtError, match="checksum"):
        load_image(p)
    open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 1)) + _chunk(b"IDAT", zlib.compress(b"\0" * 52)) + _chunk(b"IEND", b""))
    with pytest.raises(PtError, match="interlaced"):
        load_image(p)
    open(p, "wb").write(b"GIF89a")
    with pytest.raises(PtError):
        load_image(p)
    # Image::load's behaviour (src/textures/image.rs:20-25): log and use the texture without an image
    s = SceneSource()
    t = s.image_file(p)
    assert s.textures[t][-1] == -1 and "Using default texture instead" in capfd.readouterr().err
    good = str(tmp_path / "good.png")
    save_png(good, np.full((2, 3, 3), 77, np.uint8))
    t2 = s.image_file(good)
    assert s.textures[t2][-1] == 0 and s.images[0].shape == (2, 3, 3) and (s.images[0] == 77).all()


def test_a_scene_textured_from_a_file_renders_as_one_textured_from_memory(tmp_path):
    """The oracle (the checker) sees the same texels whichever way they arrive."""
    from oracle.pyoracle import OracleScene
    from pathtracer_amd import scenes
    rng = np.random.default_rng(5)
    texels = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    p = str(tmp_path / "tex.png")
    save_png(p, texels)

    def build(from_file):
        s = scenes.cornell(24, 4)
        tex = s.image_file(p) if from_file else s.image(texels)
        s.objects.add_sphere((278.0, 278.0, 278.0), 90.0, s.lambertian_texture(tex))
        return s
    a = OracleScene(build(True)).render(0, 2, seed=1)["accum"]
    b = OracleScene(build(False)).render(0, 2, seed=1)["accum"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
def test_device_renders_the_same_bits_with_texels_from_a_file(tmp_path):
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene, Renderer
    rng = np.random.default_rng(6)
    texels = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    p = str(tmp_path / "tex.png")
    save_png(p, texels)

    def render(from_file):
        s = scenes.cornell(48, 16)
        tex = s.image_file(p) if from_file else s.image(texels)
        s.objects.add_sphere((278.0, 278.0, 278.0), 120.0, s.lambertian_texture(tex))
        r = Renderer(HostScene(s))
        r.render(0, 16, seed=3)
        a = r.read_accum()[0]
        r.close()
        return a
    a, b = render(True), render(False)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0


@pytest.mark.gpu
def test_the_command_line_renderer_writes_png_and_ppm_with_the_same_pixels(tmp_path):
    """ptrender is the reference's main() without its window (src/main.rs:62-118): Cornell scene through the C++ scene
    API, Camera::render on the GPU, image saved as the reference saves it (PNG) or as PPM."""
    import os
    import subprocess
    from pathtracer_amd import scenes
    from pathtracer_amd.core import HostScene, Renderer
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracer_amd", "ptrender")
    png, ppm = str(tmp_path / "strat.png"), str(tmp_path / "strat.ppm")
    for out in (png, ppm):
        subprocess.run([exe, "64", "16", out], check=True, capture_output=True, timeout=120)
    a, b = load_image(png), load_image(ppm)
    assert a.shape == (64, 64, 3) and np.array_equal(a, b) and a.mean() > 5
    # the same scene authored in Python, the same seed: the same 8-bit image
    r = Renderer(HostScene(scenes.cornell(64, 16)))
    r.render(0, 16, seed=0)
    want = r.read_rgb8()
    r.close()
    assert np.array_equal(a, want)


def test_malformed_files_are_refused_not_crashed_on(tmp_path):
    """Valid chunk checksums around broken contents: sizes that do not match the data, unknown filters, truncated and
    over-long deflate streams, palettes that are too short, absurd headers.  Every file either loads or raises."""
    import random
    rnd = random.Random(11)
    p = str(tmp_path / "m.png")
    loaded = refused = 0
    for _ in range(400):
        color = rnd.choice([0, 2, 3, 4, 6])
        depth = rnd.choice({0: [1, 2, 4, 8, 16], 2: [8, 16], 3: [1, 2, 4, 8], 4: [8, 16], 6: [8, 16]}[color])
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        w, h = rnd.randint(1, 19), rnd.randint(1, 11)
        stride = (w * ch * depth + 7) // 8
        raw = bytearray()
        for _y in range(h):
            raw.append(rnd.randint(0, 4))
            raw += bytes(rnd.getrandbits(8) for _ in range(stride))
        defect = rnd.random()
        head = [w, h, depth, color, 0, 0, 0]
        if defect < 0.1:
            raw[rnd.randrange(0, len(raw), stride + 1)] = rnd.choice([5, 77, 255])
        elif defect < 0.2:
            raw = raw[:-rnd.randint(1, stride)]
        elif defect < 0.3:
            raw += b"\0" * rnd.randint(1, 9)
        elif defect < 0.4:
            head[rnd.randrange(0, 2)] = rnd.choice([0, 2 ** 16, 2 ** 31 - 1, w + 1])
        elif defect < 0.5:
            head[rnd.randrange(2, 7)] = rnd.choice([0, 3, 5, 7, 32, 255])
        z = zlib.compress(bytes(raw))
        if 0.5 <= defect < 0.6:
            z = z[:rnd.randint(0, len(z) - 1)]
        data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", *head))
        if color == 3:
            data += _chunk(b"PLTE", bytes(rnd.getrandbits(8) for _ in range(3 * rnd.choice([1, 2, 4, 16, 256]))))
        data += _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
        if 0.6 <= defect < 0.65:
            data = data[:rnd.randint(0, len(data) - 1)]
        open(p, "wb").write(data)
        try:
            img = load_image(p)
            assert img.shape == (h, w, 3)
            loaded += 1
        except PtError:
            refused += 1
    assert loaded > 100 and refused > 100, (loaded, refused)
