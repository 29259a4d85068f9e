"""OpenEXR scan-line files without the OpenEXR library (csrc/host/exr.cpp): what ReadImageEXR / WriteImageEXR
(src/core/imageio.cpp:138-214) exchange with the rest of pbrt. The file layout is checked against an independent
encoder / decoder written here with struct + zlib + numpy (half = numpy float16: IEEE round-to-nearest-even, as Imath's)."""
import struct
import zlib

import numpy as np
import pytest


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _zip_block(raw):
    """ImfZip: de-interleave, delta predictor, deflate; raw when that is no shorter."""
    raw = np.frombuffer(raw, np.uint8)
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 384) & 255
    packed = zlib.compress(d.astype(np.uint8).tobytes())
    return packed if len(packed) < len(raw) else raw.tobytes()


def _rle_block(raw):
    """ImfRle over the same de-interleaved, delta-coded bytes: runs of 3 .. 128 equal bytes as (n - 1, byte), other bytes in
    literal groups of up to 127 as (-n, bytes...); raw when that is no shorter."""
    raw = np.frombuffer(raw, np.uint8)
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int32)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 384) & 255
    d = d.astype(np.uint8).tobytes()
    out, i, lit = bytearray(), 0, bytearray()

    def flush():
        while lit:
            k = min(127, len(lit))
            out.extend(struct.pack("b", -k) + bytes(lit[:k]))
            del lit[:k]

    while i < len(d):
        j = i
        while j < len(d) and d[j] == d[i] and j - i < 128:
            j += 1
        if j - i >= 3:
            flush()
            out.extend(struct.pack("b", j - i - 1) + d[i:i + 1])
            i = j
        else:
            lit.append(d[i])
            i += 1
    flush()
    return bytes(out) if len(out) < len(raw) else raw.tobytes()


def _unzip_block(data, n):
    if len(data) == n:
        return np.frombuffer(data, np.uint8)
    d = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int64)
    assert len(d) == n
    t = (np.cumsum(d - 128) + 128) & 255  # t[i] = t[i-1] + d[i] - 128, t[0] = d[0]
    t = t.astype(np.uint8)
    half = (n + 1) // 2
    out = np.empty(n, np.uint8)
    out[0::2], out[1::2] = t[:half], t[half:]
    return out


def _make_exr(channels, data_window, compression, line_order=0, display=None, version=2):
    """channels: list of (name, type 0 UINT / 1 HALF / 2 FLOAT, (h, w) array), stored alphabetically."""
    channels = sorted(channels, key=lambda c: c[0])
    x0, y0, x1, y1 = data_window
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chl = b""
    for name, typ, _ in channels:
        chl += name.encode() + b"\0" + struct.pack("<iBBBBii", typ, 0, 0, 0, 0, 1, 1)
    chl += b"\0"
    dx0, dy0, dx1, dy1 = display or data_window
    head = struct.pack("<II", 0x01312F76, version)
    head += _attr("channels", "chlist", chl)
    head += _attr("compression", "compression", bytes([compression]))
    head += _attr("dataWindow", "box2i", struct.pack("<4i", x0, y0, x1, y1))
    head += _attr("displayWindow", "box2i", struct.pack("<4i", dx0, dy0, dx1, dy1))
    head += _attr("lineOrder", "lineOrder", bytes([line_order]))
    head += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    per = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
    blocks = []
    for first in range(0, h, per):
        raw = b""
        for y in range(first, min(first + per, h)):
            for _, typ, arr in channels:
                raw += arr[y].astype({0: "<u4", 1: "<f2", 2: "<f4"}[typ]).tobytes()
        blocks.append((y0 + first, raw if compression == 0 else _rle_block(raw) if compression == 1 else _zip_block(raw)))
    order = blocks if line_order == 0 else blocks[::-1]  # decreasing Y: chunks stored bottom-up, table still by block
    pos = len(head) + 8 * len(blocks)
    offsets, body = {}, b""
    for y, data in order:
        offsets[y] = pos + len(body)
        body += struct.pack("<ii", y, len(data)) + data
    table = b"".join(struct.pack("<Q", offsets[y]) for y, _ in blocks)
    return head + table + body


def _as_half(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def test_half_conversion_is_numpys(binding, tmp_path):
    rng = np.random.default_rng(7)
    vals = np.concatenate([
        np.array([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, np.inf, -np.inf, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8,
                  2.9802322e-8, 2.99e-8, 1e-10, 0.1, 1 / 3, 2049.0, 2051.0, 1.00048828125, 1.00146484375], np.float32),
        rng.standard_normal(4000).astype(np.float32) * 100, (2.0 ** rng.uniform(-30, 17, 4000)).astype(np.float32),
        np.float32(2.0) ** np.arange(-26, 17, dtype=np.float32) * np.float32(1.0009765625 - 2 ** -12)])  # just below ties
    n = len(vals) - len(vals) % 3
    img = vals[:n].reshape(1, -1, 3)
    binding.write_exr(str(tmp_path / "h.exr"), img)
    back = binding.read_image(str(tmp_path / "h.exr"))
    with np.errstate(over="ignore"):
        assert np.array_equal(back.view(np.uint32), _as_half(img).view(np.uint32))
    nan = np.full((1, 1, 3), np.nan, np.float32)
    binding.write_exr(str(tmp_path / "n.exr"), nan)
    assert np.isnan(binding.read_image(str(tmp_path / "n.exr"))).all()


def test_written_file_decodes_independently(binding, tmp_path):
    rng = np.random.default_rng(3)
    img = (rng.random((41, 29, 3)) * 4).astype(np.float32)
    path = tmp_path / "w.exr"
    binding.write_exr(str(path), img, origin=(5, 7), display=(64, 80))
    d = path.read_bytes()
    assert struct.unpack_from("<II", d, 0) == (0x01312F76, 2)
    pos, attrs = 8, {}
    while d[pos] != 0:
        e = d.index(b"\0", pos)
        name = d[pos:e].decode()
        e2 = d.index(b"\0", e + 1)
        size = struct.unpack_from("<i", d, e2 + 1)[0]
        attrs[name] = (d[e + 1:e2].decode(), d[e2 + 5:e2 + 5 + size])
        pos = e2 + 5 + size
    pos += 1
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (5, 7, 33, 47) and struct.unpack("<4i", attrs["displayWindow"][1]) == (0, 0, 63, 79)
    assert attrs["compression"][1] == b"\x03" and attrs["lineOrder"][1] == b"\x00"
    names = [c.split(b"\0")[0] for c in (attrs["channels"][1][0:18], attrs["channels"][1][18:36], attrs["channels"][1][36:54])]
    assert names == [b"B", b"G", b"R"]
    for req in ("pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"):
        assert req in attrs
    n_blocks = (41 + 15) // 16
    offs = struct.unpack_from("<%dQ" % n_blocks, d, pos)
    out = np.zeros((41, 29, 3), np.float32)
    for b, off in enumerate(offs):
        y, size = struct.unpack_from("<ii", d, off)
        assert y == 7 + 16 * b
        lines = min(16, 41 - 16 * b)
        raw = _unzip_block(d[off + 8:off + 8 + size], lines * 29 * 6).tobytes()
        blk = np.frombuffer(raw, "<f2").reshape(lines, 3, 29).astype(np.float32)  # per line: B, G, R
        out[16 * b:16 * b + lines] = blk[:, ::-1, :].transpose(0, 2, 1)
    assert np.array_equal(out, _as_half(img))
    assert np.array_equal(binding.read_image(str(path)), _as_half(img))


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("line_order", [0, 1])
def test_reads_independently_encoded_files(binding, tmp_path, compression, line_order):
    """FLOAT and HALF channels mixed, an alpha channel to skip, both line orders, every supported coder: values arrive as
    halfs (Imf::RgbaInputFile), the image is the data window."""
    rng = np.random.default_rng(11 + compression)
    h, w = 37, 23
    r, g, b, a = (rng.random((h, w)).astype(np.float32) * s for s in (1, 100, 1e-3, 1))
    path = tmp_path / "f.exr"
    path.write_bytes(_make_exr([("R", 2, r), ("G", 1, g), ("B", 2, b), ("A", 1, a)], (3, 4, 3 + w - 1, 4 + h - 1), compression, line_order,
                               display=(0, 0, 99, 99)))
    if compression == 1:  # something for the run-length coder to find
        g[5:9] = 0.5
        a[:] = 1.0
        path.write_bytes(_make_exr([("R", 2, r), ("G", 1, g), ("B", 2, b), ("A", 1, a)], (3, 4, 3 + w - 1, 4 + h - 1), compression, line_order,
                                   display=(0, 0, 99, 99)))
    img = binding.read_image(str(path))
    assert img.shape == (h, w, 3)
    assert np.array_equal(img, np.stack([_as_half(r), _as_half(g), _as_half(b)], axis=-1))


def test_luminance_uint_and_refusals(binding, tmp_path):
    y = np.arange(12, dtype=np.float32).reshape(3, 4) / 7
    (tmp_path / "y.exr").write_bytes(_make_exr([("Y", 1, y)], (0, 0, 3, 2), 0))
    img = binding.read_image(str(tmp_path / "y.exr"))
    assert np.array_equal(img, np.repeat(_as_half(y)[..., None], 3, axis=-1))
    u = np.array([[0, 1, 2048, 2049, 70000]], np.uint32)
    (tmp_path / "u.exr").write_bytes(_make_exr([("R", 0, u)], (0, 0, 4, 0), 0))
    with np.errstate(over="ignore"):
        want = u.astype(np.float32).astype(np.float16).astype(np.float32)
    got = binding.read_image(str(tmp_path / "u.exr"))
    assert np.array_equal(got[..., 0], want) and not got[..., 1:].any()
    good = _make_exr([("R", 1, y)], (0, 0, 3, 2), 0)
    piz = good.replace(_attr("compression", "compression", b"\0"), _attr("compression", "compression", b"\x04"))
    (tmp_path / "p.exr").write_bytes(piz)
    with pytest.raises(RuntimeError, match="PIZ"):
        binding.read_image(str(tmp_path / "p.exr"))
    (tmp_path / "t.exr").write_bytes(_make_exr([("R", 1, y)], (0, 0, 3, 2), 0, version=2 | 0x200))
    with pytest.raises(RuntimeError, match="tiled"):
        binding.read_image(str(tmp_path / "t.exr"))
    (tmp_path / "s.exr").write_bytes(good[:len(good) - 5])
    with pytest.raises(RuntimeError, match="truncated|beyond"):
        binding.read_image(str(tmp_path / "s.exr"))
    (tmp_path / "z.exr").write_bytes(b"not an exr at all")
    with pytest.raises(RuntimeError, match="OpenEXR"):
        binding.read_image(str(tmp_path / "z.exr"))


def test_film_written_as_the_scene_asks(binding, tmp_path):
    """Film::WriteImage: the shipped scene names an .exr; the cropped pixel bounds become the data window."""
    scene = binding.HostScene(xres=40, yres=30, spp=1)
    assert scene.film_filename == "killeroo-simple.exr"
    rgb = np.random.default_rng(5).random((30, 40, 3)).astype(np.float32)
    out = tmp_path / scene.film_filename
    scene.write_image(str(out), rgb)
    assert np.array_equal(binding.read_image(str(out)), _as_half(rgb))
    with pytest.raises(RuntimeError, match="suffix"):
        scene.write_image(str(tmp_path / "x.jpg"), rgb)
