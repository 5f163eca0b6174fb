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


def _pair_from_clean(clean, seed, height, width, amp_left=3, amp_right=2):
    """(left, right) from a noise-free left image: left = clean + U[-amp_left, amp_left]; right = clean shifted by the row-dependent integer
    disparity d(y) = 6 + floor(24 y / H) (edge replicated) + U[-amp_right, amp_right]."""
    left = np.clip(clean + (_noise(seed, 3, height, width, amp_left) if amp_left else 0), 0, 255).astype(np.uint8)
    d = 6 + (24 * np.arange(height)) // height
    cols = np.minimum(np.arange(width)[None, :] + d[:, None], width - 1)
    shifted = np.take_along_axis(clean, cols, axis=1)
    right = np.clip(shifted + (_noise(seed, 4, height, width, amp_right) if amp_right else 0), 0, 255).astype(np.uint8)
    return left, right


def synth_stereo_pair(seed, height=480, width=752):
    """Return (left, right) uint8 images of shape (height, width)."""
    return _pair_from_clean(_clean_left(seed, height, width), seed, height, width)


# ---- other image statistics (bench.py `other_inputs`, tests): the kernels' list / pool / spill sizes were tuned on synth_stereo_pair only ----
INPUT_FAMILIES = ("noise", "saltpepper", "lowtexture", "natural")


def _value_noise(seed, stream, height, width, octave):
    """integer value noise: U[0, 255] on a lattice of spacing 2^octave, bilinearly interpolated with exact integer weights; returns values * 4^octave"""
    s = 1 << octave
    gh, gw = height // s + 2, width // s + 2
    lat = _uniform_int(_stream(seed, stream, gh * gw), 0, 255).reshape(gh, gw)
    y, x = np.arange(height), np.arange(width)
    y0, fy = y >> octave, y & (s - 1)
    x0, fx = x >> octave, x & (s - 1)
    a = lat[y0][:, x0] * (s - fx)[None, :] + lat[y0][:, x0 + 1] * fx[None, :]
    b = lat[y0 + 1][:, x0] * (s - fx)[None, :] + lat[y0 + 1][:, x0 + 1] * fx[None, :]
    return a * (s - fy)[:, None] + b * fy[:, None]


def synth_family_pair(kind, seed, height=480, width=752):
    """Stereo pairs with other statistics than synth_stereo_pair (same disparity model, integer arithmetic only - reproducible anywhere):
      noise       every pixel U[0, 255]: nearly every pixel passes the early rejects, a fifth of them are FAST corners - every band of k_detect spills
      saltpepper  mid grey 128 with 25 % of the pixels set to 0 or 255 (isolated extreme pixels: the densest corner field an image can hold)
      lowtexture  flat grey + U[-1, 1] noise with half a dozen rectangles: ~100 keypoints per image, nearly every tile empty
      natural     a 1/f-like field: seven octaves of integer value noise, amplitude proportional to the lattice spacing, + U[-2, 2] noise"""
    if kind == "noise":
        clean = _uniform_int(_stream(seed, 11, height * width), 0, 255).astype(np.int16).reshape(height, width)
        return _pair_from_clean(clean, seed, height, width, 0, 2)
    if kind == "saltpepper":
        u = _stream(seed, 12, height * width)
        sel = (u % np.uint64(8)).reshape(height, width)
        clean = np.full((height, width), 128, np.int16)
        clean[sel == 0] = 0
        clean[sel == 1] = 255
        return _pair_from_clean(clean, seed, height, width, 2, 2)
    if kind == "lowtexture":
        clean = np.full((height, width), 120, np.int16)
        area = height * width
        n_rect = max(3, int(round(6.0 * area / (752.0 * 480.0))))
        u = _stream(seed, 13, 5 * n_rect)
        rw, rh = _uniform_int(u[0::5], 20, 90), _uniform_int(u[1::5], 20, 90)
        rx, ry = _uniform_int(u[2::5], 30, max(30, width - 120)), _uniform_int(u[3::5], 30, max(30, height - 120))
        rv = _uniform_int(u[4::5], 0, 255)
        for k in range(n_rect):
            clean[int(ry[k]):int(ry[k] + rh[k]), int(rx[k]):int(rx[k] + rw[k])] = rv[k]
        return _pair_from_clean(clean, seed, height, width, 1, 1)
    if kind == "natural":
        acc = np.zeros((height, width), np.int64)
        wsum = 0
        for o in range(7):
            # amplitude ~ lattice spacing (1/f): value noise of octave o comes scaled by 4^o; bring every octave to a common 2^12 scale, weight 2^o
            acc += (_value_noise(seed, 20 + o, height, width, o) << (12 - 2 * o)) * (1 << o)
            wsum += 1 << o
        mean = (acc // wsum) >> 12                                                 # 0..255, heavily averaged towards 128
        clean = np.clip((mean - 128) * 3 + 128, 0, 255).astype(np.int16)            # stretch the contrast back
        return _pair_from_clean(clean, seed, height, width, 2, 2)
    raise ValueError("unknown input family %r" % (kind,))


def synth_image(seed, height=480, width=752):
    return synth_stereo_pair(seed, height, width)[0]


def synth_adversarial_pair(seed, height=480, width=752):
    """A stereo pair whose RIGHT image is built by row bands so that the rarely taken branches of the stereo matcher's tail are exercised
    (tools/ptx_chain.py chain `i`, tests/test_ptx_chain.py; every synthetic pair above has a positive disparity of 6-30 px everywhere):
      band 0  rows [0, 5H/24)      right == left bit for bit, no noise, plus mirror-symmetric features (ends of thin vertical bars): zero disparity, and
                                   where the L1 sums left and right of the minimum tie, a disparity of exactly 0 -> the `disparity <= 0 -> 0.01` branch,
                                   the one double-precision expression of the path (orb_stereo_match.cu:538-545)
      band 1  rows [5H/24, 9H/24)  content shifted the WRONG way (disparity -7): candidates outside [uL - maxD, uL], wrong or no matches
      band 2  rows [9H/24, 13H/24) disparities 19..22, around maxD when the chain's fx is 20 (maxD = mbf / mb = fx): candidates inside the window whose
                                   refined disparity ends up on either side of maxD
      band 3a rows [13H/24, 16H/24) no noise, right = mean of the left content shifted by 8 and by 9 px (a half-pixel disparity): the L1 sums of the two
                                   neighbouring shifts tie exactly (deltaR = +-0.5)
      band 3b rows [16H/24, 19H/24) ordinary disparities 4..16 with noise (so that the median cut has a population)
      band 4  rows [19H/24, H)     a comb of identical thin bars with period 12 and disparity 3: the best ORB candidate is often a whole period away,
                                   the L1 minimum then sits on the edge of the +-5 window (bestR == 0 / 10 -> rejected) or ties with its neighbour
    Returns (left, right) uint8."""
    clean = _clean_left(seed, height, width)
    b = [0, 5 * height // 24, 9 * height // 24, 13 * height // 24, 19 * height // 24, height]
    # band 0: symmetric bar ends on a light background patch, dark bars of width 3 / 5, every 37 px, rows chosen per bar
    u = _stream(seed, 7, 4 * 64)
    y_lo, y_hi = b[0] + 24, b[1] - 30
    k = 0
    for x0 in range(30, width - 30, 37):
        wbar = 3 if (k & 1) else 5
        ytop = int(_uniform_int(u[k:k + 1], y_lo, max(y_lo, y_hi - 20))[0])
        clean[max(0, ytop - 12):ytop + 26, x0 - 12:x0 + 12 + wbar] = 200
        clean[ytop:ytop + 26, x0:x0 + wbar] = 30
        k += 1
    # band 4: comb
    yb = b[4]
    clean[yb:, :] = 150
    for x0 in range(24, width - 24, 12):
        top = yb + 22 + 14 * ((x0 // 12) % 3)
        clean[top:min(height, top + 40), x0:x0 + 3] = 40
    left = np.clip(clean + _noise(seed, 3, height, width, 3), 0, 255).astype(np.uint8)
    left[b[0]:b[1]] = np.clip(clean[b[0]:b[1]], 0, 255).astype(np.uint8)          # band 0: no noise
    b3a = 16 * height // 24
    left[b[3]:b3a] = np.clip(clean[b[3]:b3a], 0, 255).astype(np.uint8)            # band 3a: no noise
    d = np.zeros(height, np.int64)
    d[b[1]:b[2]] = -7
    d[b[2]:b[3]] = 19 + (np.arange(b[2], b[3]) // 5) % 4
    d[b[3]:b3a] = 8
    rows3 = np.arange(b3a, b[4])
    d[b3a:b[4]] = 4 + (12 * (rows3 - b3a)) // max(1, b[4] - b3a)
    d[b[4]:] = 3
    cols = np.clip(np.arange(width)[None, :] + d[:, None], 0, width - 1)
    shifted = np.take_along_axis(clean, cols, axis=1)
    right = np.clip(shifted + _noise(seed, 4, height, width, 2), 0, 255).astype(np.uint8)
    right[b[0]:b[1]] = left[b[0]:b[1]]
    cols9 = np.clip(np.arange(width)[None, :] + 9, 0, width - 1) + np.zeros((b3a - b[3], 1), np.int64)
    right[b[3]:b3a] = np.clip((shifted[b[3]:b3a] + np.take_along_axis(clean[b[3]:b3a], cols9, axis=1) + 1) // 2, 0, 255).astype(np.uint8)
    return left, right
