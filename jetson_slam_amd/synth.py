"""Deterministic, seed-addressed synthetic stereo pairs.

No dataset can be downloaded here, so benchmarks and parity tests use EuRoC/KITTI-shaped synthetic
images: mid-grey background + random axis-aligned rectangles + 8x8 checker patches + small uniform
noise.  The right image is the (noise-free) left image shifted by an integer, row-dependent
disparity d(y) = 6 + floor(24*y/H) with edge replication, plus independent noise.
Everything is derived from a counter-based splitmix64 so that any (seed, pair) is reproducible
bit-for-bit on any machine / numpy version.
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(x):
    with np.errstate(over="ignore"):
        z = x + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _stream(seed, stream, n):
    """n uint64 values of stream `stream` of generator `seed` (counter based)."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([seed & 0xFFFFFFFFFFFFFFFF], np.uint64) ^ (np.uint64(stream) * _M2))
        return _mix(base + np.arange(n, dtype=np.uint64) * _M1)


def _uniform_int(u, lo, hi):
    """map uint64 -> integers in [lo, hi] (the tiny modulo bias is irrelevant here)."""
    return (u % np.uint64(hi - lo + 1)).astype(np.int64) + lo


def _clean_left(seed, height, width):
    img = np.full((height, width), 128, np.int16)
    area = height * width
    n_rect = max(24, int(round(600.0 * area / (752.0 * 480.0))))
    u = _stream(seed, 1, 5 * n_rect)
    rw = _uniform_int(u[0::5], 6, 60)
    rh = _uniform_int(u[1::5], 6, 60)
    rx = _uniform_int(u[2::5], -20, width - 1)
    ry = _uniform_int(u[3::5], -20, height - 1)
    rv = _uniform_int(u[4::5], 0, 255)
    for k in range(n_rect):
        x0, y0 = max(0, int(rx[k])), max(0, int(ry[k]))
        x1, y1 = min(width, int(rx[k] + rw[k])), min(height, int(ry[k] + rh[k]))
        if x1 > x0 and y1 > y0:
            img[y0:y1, x0:x1] = rv[k]
    n_chk = max(4, int(round(40.0 * area / (752.0 * 480.0))))
    u = _stream(seed, 2, 4 * n_chk)
    cx = _uniform_int(u[0::4], 0, max(0, width - 33))
    cy = _uniform_int(u[1::4], 0, max(0, height - 33))
    ca = _uniform_int(u[2::4], 0, 255)
    cb = _uniform_int(u[3::4], 0, 255)
    yy, xx = np.mgrid[0:32, 0:32]
    chk = ((yy // 8 + xx // 8) & 1).astype(bool)
    for k in range(n_chk):
        x0, y0 = int(cx[k]), int(cy[k])
        h, w = min(32, height - y0), min(32, width - x0)
        img[y0:y0 + h, x0:x0 + w] = np.where(chk[:h, :w], ca[k], cb[k])
    return img


def _noise(seed, stream, height, width, amp):
    u = _stream(seed, stream, height * width)
    return _uniform_int(u, -amp, amp).astype(np.int16).reshape(height, width)


def synth_stereo_pair(seed, height=480, width=752):
    """Return (left, right) uint8 images of shape (height, width)."""
    clean = _clean_left(seed, height, width)
    left = np.clip(clean + _noise(seed, 3, height, width, 3), 0, 255).astype(np.uint8)
    d = 6 + (24 * np.arange(height)) // height
    cols = np.minimum(np.arange(width)[None, :] + d[:, None], width - 1)
    shifted = np.take_along_axis(clean, cols, axis=1)
    right = np.clip(shifted + _noise(seed, 4, height, width, 2), 0, 255).astype(np.uint8)
    return left, right


def synth_image(seed, height=480, width=752):
    return synth_stereo_pair(seed, height, width)[0]
