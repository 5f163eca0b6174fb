"""jsorb_read_mask_image: the mask file the reference loads with cv::imread + cvtColor(BGR2GRAY) (orb_gpu.cpp:64-75), decoded without
OpenCV.  Host-only code, so it runs on CPU: PNGs are written here (zlib + hand-made chunks, every colour type / bit depth / scanline
filter the decoder claims, stored / fixed / dynamic DEFLATE blocks) and compared with the gray plane computed in numpy."""
import os
import struct
import zlib

import numpy as np
import pytest


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(rows, bpp, filters):
    """rows: list of bytes per scanline -> filtered stream with the given filter type per row"""
    out = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        out.append(ft)
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][ft]
            out.append((v - pred) & 0xFF)
        prev = row
    return bytes(out)


def _write_png(path, w, h, depth, ctype, rows, bpp, filters=(0, 1, 2, 3, 4), level=6, plte=None, idat_split=1, extra_chunk=True):
    raw = _filter_rows(rows, bpp, filters)
    comp = zlib.compress(raw, level)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        if extra_chunk:
            f.write(_chunk(b"tEXt", b"Comment\x00jsorb test"))
        if plte is not None:
            f.write(_chunk(b"PLTE", bytes(plte)))
        step = max(1, len(comp) // idat_split)
        for i in range(0, len(comp), step):
            f.write(_chunk(b"IDAT", comp[i:i + step]))
        f.write(_chunk(b"IEND", b""))


def _gray_of_rgb(rgb):
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


@pytest.fixture(scope="module")
def orb_mod():
    import __graft_entry__ as g
    g.build()
    from jetson_slam_amd import orb
    return orb


@pytest.mark.parametrize("level", [0, 1, 9])                # stored, fast (often fixed Huffman on tiny inputs), dynamic Huffman
def test_png_gray_rgb_rgba_palette_16bit_and_filters(orb_mod, tmp_path, level):
    rng = np.random.default_rng(5 + level)
    w, h = 67, 41
    # smooth + noisy content so that the filters and the LZ77 matches all get exercised
    yy, xx = np.mgrid[0:h, 0:w]
    base = ((xx * 3 + yy * 5) % 256).astype(np.uint8)
    gray = np.where(rng.random((h, w)) < 0.2, rng.integers(0, 256, (h, w)), base).astype(np.uint8)
    rgb = np.stack([gray, np.roll(gray, 3, 1), 255 - gray], -1)
    alpha = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
    cases = {
        "gray8": (8, 0, [gray[y].tobytes() for y in range(h)], 1, gray),
        "ga8": (8, 4, [np.concatenate([gray[y][:, None], alpha[y]], 1).tobytes() for y in range(h)], 2, gray),
        "rgb8": (8, 2, [rgb[y].tobytes() for y in range(h)], 3, _gray_of_rgb(rgb)),
        "rgba8": (8, 6, [np.concatenate([rgb[y], alpha[y]], 1).tobytes() for y in range(h)], 4, _gray_of_rgb(rgb)),
        "gray16": (16, 0, [np.stack([gray[y], alpha[y, :, 0]], 1).tobytes() for y in range(h)], 2, gray),        # high byte first
        "rgb16": (16, 2, [np.stack([rgb[y, :, 0], alpha[y, :, 0], rgb[y, :, 1], alpha[y, :, 0], rgb[y, :, 2], alpha[y, :, 0]], 1).tobytes() for y in range(h)], 6,
                  _gray_of_rgb(rgb)),
    }
    for name, (depth, ctype, rows, bpp, want) in cases.items():
        p = str(tmp_path / (name + ".png"))
        _write_png(p, w, h, depth, ctype, rows, bpp, level=level, idat_split=3)
        got = orb_mod.read_mask_image(p)
        assert got is not None and got.shape == (h, w) and np.array_equal(got, want), (name, level)
    # palette, 8 and 4 bit
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (h, w), dtype=np.uint8)
    want = _gray_of_rgb(pal[idx])
    p = str(tmp_path / "pal8.png")
    _write_png(p, w, h, 8, 3, [idx[y].tobytes() for y in range(h)], 1, level=level, plte=pal.reshape(-1))
    assert np.array_equal(orb_mod.read_mask_image(p), want)
    rows4 = []
    for y in range(h):
        r = np.concatenate([idx[y], np.zeros(w % 2, np.uint8)])
        rows4.append(((r[0::2] << 4) | r[1::2]).astype(np.uint8).tobytes())
    p = str(tmp_path / "pal4.png")
    _write_png(p, w, h, 4, 3, rows4, 1, level=level, plte=pal.reshape(-1))
    assert np.array_equal(orb_mod.read_mask_image(p), want)
    # 1-bit gray (a typical hand-drawn mask): expanded to 0 / 255
    bits = (rng.random((h, w)) < 0.5).astype(np.uint8)
    rows1 = [np.packbits(bits[y]).tobytes() for y in range(h)]
    p = str(tmp_path / "gray1.png")
    _write_png(p, w, h, 1, 0, rows1, 1, level=level)
    assert np.array_equal(orb_mod.read_mask_image(p), bits * 255)


def test_pnm_and_error_cases(orb_mod, tmp_path):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (13, 21), dtype=np.uint8)
    p = str(tmp_path / "m.pgm")
    open(p, "wb").write(b"P5\n# a comment\n21 13\n255\n" + g.tobytes())
    assert np.array_equal(orb_mod.read_mask_image(p), g)
    rgb = rng.integers(0, 256, (13, 21, 3), dtype=np.uint8)
    p = str(tmp_path / "m.ppm")
    open(p, "wb").write(b"P6 21 13 255\n" + rgb.tobytes())
    assert np.array_equal(orb_mod.read_mask_image(p), _gray_of_rgb(rgb))
    assert orb_mod.read_mask_image(str(tmp_path / "missing.png")) is None          # unreadable = "no mask", as in the reference
    p = str(tmp_path / "bad.png")
    open(p, "wb").write(b"this is not an image at all, just some bytes")
    with pytest.raises(orb_mod.JsorbError):
        orb_mod.read_mask_image(p)
    # truncated zlib stream
    p = str(tmp_path / "trunc.png")
    _write_png(p, 8, 8, 8, 0, [bytes(8)] * 8, 1)
    data = open(p, "rb").read()
    open(p, "wb").write(data[:len(data) - 30])
    with pytest.raises(orb_mod.JsorbError):
        orb_mod.read_mask_image(p)


@pytest.mark.gpu
def test_mask_by_file_name_like_the_reference_constructor(orb_mod, po, tmp_path):
    """ORBExtractor(..., str_mask = "<file>.png", ...) - the reference's constructor argument (Tracking.cpp:180-216): the PNG is decoded,
    resized to the level-0 size with the INTER_NN index rule when it has another size, and thresholded per level."""
    from jetson_slam_amd.synth import synth_stereo_pair
    H, W = 240, 320
    mask = np.full((H // 2, W // 2), 255, np.uint8)         # half-size mask: exercises the resize to level 0
    mask[30:90, 50:130] = 0
    p = str(tmp_path / "mask.png")
    _write_png(p, W // 2, H // 2, 8, 0, [mask[y].tobytes() for y in range(H // 2)], 1)
    full = mask[np.minimum((np.arange(H) * (1.0 / (H / float(H // 2)))).astype(int), H // 2 - 1)][:, np.minimum((np.arange(W) * (1.0 / (W / float(W // 2)))).astype(int), W // 2 - 1)]
    g = orb_mod.ORBExtractor(H, W, 1.2, 3, 9, 14, 7, 20, p, 15, 15)
    o = po.OracleExtractor(height=H, width=W, n_levels=3, tile_h=15, tile_w=15, mask=np.ascontiguousarray(full))
    img, _ = synth_stereo_pair(61, H, W)
    kp, desc = g.extract(img)
    o.extract(img)
    assert np.array_equal(kp, o.keypoints()) and np.array_equal(desc, o.descriptors())
    for lv in range(3):
        assert np.array_equal(g.level_mask(lv), o.level_mask(lv))
    assert len(kp) // 6 > 50
