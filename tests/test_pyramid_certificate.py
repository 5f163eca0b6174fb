"""k_pyramid's certified fast path (jetson_slam_amd/csrc/k_pyramid.hip): A = wyt*T + wyb*B with T = fma(wxl, TL, wxr*TR) (and B on the
row below) decides trunc(C) of the reference's chain C = (wxr*wyt)*TR, fma(wxl*wyt, TL), fma(wxl*wyb, BL), fma(wxr*wyb, BR) whenever
floor(256 A) mod 256 is neither 0 nor 255, i.e. whenever A is farther than 2^-8 from an integer; the other pixels go through the
chain.  Where wxr == 0 or wyb == 0 the chain degenerates to A's own operations and A is the chain's value bit for bit (no
certificate).  This test derives the rigorous bound on |A - C| with exact rational arithmetic, requires the 2^-8 band to cover it
with margin, and checks the decision rule (as the kernel evaluates it: one addition of 49152.0f rounded down, bytes of the mantissa)
and the degenerate cases on adversarial inputs against the chain (the bit-exact -m gpu plane comparisons finally guard the kernel)."""
import os
import re
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = Fraction(1, 2 ** 24)          # unit roundoff of f32


def _magic():
    src = open(os.path.join(ROOT, "jetson_slam_amd", "csrc", "k_pyramid.hip")).read()
    m = re.search(r"magic = \(pyr_f2\)\{([0-9.]+)f, ", src)
    return float(m.group(1))


def test_band_covers_the_rigorous_error_bound():
    """Both values approximate S = sum of (weight product as a real number) * pixel.  Weights lie in [0, 1], wxl + wxr and wyt + wyb
    are 1 up to one rounding of the subtraction (1 - w), pixels are <= 255."""
    one = 1 + U                                   # wxl + wxr <= 1 + u (wxr = fl(1 - wxl))
    smax = 255 * one * one                        # bound on S and on every partial sum
    # reference chain: 4 rounded weight products (relative u each) + 4 roundings of partial sums
    bound_c = U * smax + 4 * U * smax * (1 + U) ** 4
    # fast path: T, B = one rounded product + one fma (2 roundings, values <= 255 (1+u)); A = one rounded product + one fma on them
    tmax = 255 * one
    bound_t = 2 * U * tmax * (1 + U)
    bound_a = one * bound_t + 2 * U * (smax + bound_t) * (1 + U)
    total = bound_c + bound_a
    assert float(total) < 1.5e-4, float(total)
    magic = _magic()
    assert magic == 49152.0                        # 1.5 * 2^15: ulp 2^-8 over [32768, 65536), low 16 mantissa bits = round(256 A)
    band = Fraction(1, 256)                        # fraction byte in {0, 255}  <=>  A within 2^-8 of an integer
    assert float(band) > 20 * float(total)         # the undecided band is more than an order of magnitude wider than the bound


def _chain(wxl, wxr, wyt, wyb, tl, tr, bl, br):
    f32, f64 = np.float32, np.float64
    acc = ((wxr * wyt).astype(f32) * tr).astype(f32)
    for w, p in (((wxl * wyt).astype(f32), tl), ((wxl * wyb).astype(f32), bl), ((wxr * wyb).astype(f32), br)):
        acc = (w.astype(f64) * p.astype(f64) + acc.astype(f64)).astype(f32)        # fma: exact product (f32 x u8 fits f64), one rounding
    return acc


def _fast(wxl, wxr, wyt, wyb, tl, tr, bl, br):
    f32, f64 = np.float32, np.float64
    t = (wxl.astype(f64) * tl + (wxr * tr).astype(f32).astype(f64)).astype(f32)
    b = (wxl.astype(f64) * bl + (wxr * br).astype(f32).astype(f64)).astype(f32)
    return (wyb.astype(f64) * b.astype(f64) + (wyt * t).astype(f32).astype(f64)).astype(f32)


def test_decision_rule_never_lies_and_decides_almost_everything():
    rng = np.random.default_rng(7)
    n = 400000
    f32 = np.float32
    magic = f32(_magic())

    def weights(k):
        # the reference's weights: wxl = (xl + 1) - fx for fx = s * w, wxr = 1 - wxl; include exact 0 / 1 and tiny values
        s = f32(1.2) ** rng.integers(1, 8, k).astype(f32)
        fx = (s * rng.integers(0, 2000, k).astype(f32)).astype(f32)
        wl = (np.floor(fx) + f32(1.0) - fx).astype(f32)
        return wl, (f32(1.0) - wl).astype(f32)

    cases = []
    wxl, wxr = weights(n); wyt, wyb = weights(n)
    px = lambda lo, hi: [rng.integers(lo, hi, n).astype(f32) for _ in range(4)]
    cases.append((wxl, wxr, wyt, wyb, *px(0, 256)))                                 # generic
    v = rng.integers(0, 256, n).astype(f32)
    cases.append((wxl, wxr, wyt, wyb, v, v, v, v))                                   # flat 2x2 blocks: the real value is an integer
    cases.append((wxl, wxr, wyt, wyb, *px(254, 256)))                                # near saturation
    cases.append((wxl, wxr, wyt, wyb, v, v, np.minimum(v + 1, 255), np.minimum(v + 1, 255)))   # vertical unit steps
    one, zero = np.ones(n, f32), np.zeros(n, f32)
    cases.append((one, zero, one, zero, *px(0, 256)))                                # column 0 / row 0: A = TL exactly
    cases.append((f32(0.5) * one, f32(0.5) * one, f32(0.5) * one, f32(0.5) * one, *px(0, 256)))   # quarters
    decided = total = 0
    for c in cases:
        C, A = _chain(*c), _fast(*c)
        assert np.abs(A.astype(np.float64) - C.astype(np.float64)).max() < 2e-4
        q = np.floor(A.astype(np.float64) * 256.0).astype(np.int64)                # A + 49152 rounded down: low 16 mantissa bits = floor(256 A)
        frac, ipart = q & 0xFF, q >> 8
        sure = (frac != 0) & (frac != 255)
        assert np.array_equal(ipart[sure], np.floor(C[sure]).astype(np.int64))      # a decided pixel is always the reference's byte
        exact = (c[1] == 0) | (c[3] == 0)                                           # wxr == 0 or wyb == 0: no certificate, A must BE the chain
        assert np.array_equal(A[exact].view(np.uint32), C[exact].view(np.uint32))
        assert np.array_equal(ipart[exact], np.floor(C[exact]).astype(np.int64))
        decided += int((sure | exact).sum()); total += len(sure)
    assert decided > 0.6 * total
    C, A = _chain(*cases[0]), _fast(*cases[0])
    q = np.floor(A.astype(np.float64) * 256.0).astype(np.int64) & 0xFF
    assert ((q != 0) & (q != 255)).mean() > 0.85                                    # (scale-1.2 weights put many values on a 1/25 lattice)


def test_scale_1p2_level_1_weights_are_often_exactly_zero_every_fifth_column():
    """s * w is an exact f32 integer for most w = 0, 5, 10, ... at scale 1.2 (the product of the f32 nearest to 1.2 and a multiple of 5
    is a tie that rounds to the integer about 70 % of the time) - the case the kernel handles without a certificate.  Where it is not,
    the pixels are certified or listed like any other: this only documents why the shortcut is worth having."""
    f32 = np.float32
    s = f32(1.0) / (f32(1.0) / f32(1.2))
    w = np.arange(0, 640, dtype=f32)
    fx = (s * w).astype(f32)
    wxr = (f32(1.0) - ((np.floor(fx) + f32(1.0)) - fx).astype(f32)).astype(f32)
    assert (wxr[::5] == 0).mean() > 0.5 and np.all(wxr[np.arange(640) % 5 != 0] != 0)


def test_zero_byte_flags_have_no_false_negative():
    """y = (f ^ f << 1) & 0xFEFEFEFE ; z = ~y & (y - 0x01010101): bit 8t+7 of z must be set whenever byte t of f is 0 or 255 (false
    positives only cost an exact recomputation)."""
    rng = np.random.default_rng(3)
    f = rng.integers(0, 2 ** 32, 500000, dtype=np.uint64).astype(np.uint32)
    f[::3] &= np.uint32(0x00FF01FF); f[1::5] |= np.uint32(0xFF0000FF); f[2::7] &= np.uint32(0xFF0000FF); f[:16] = 0; f[16:32] = 0xFFFFFFFF
    y = (f ^ (f << np.uint32(1))) & np.uint32(0xFEFEFEFE)
    z = ~y & (y - np.uint32(0x01010101))
    for t in range(4):
        byte = (f >> np.uint32(8 * t)) & np.uint32(0xFF)
        hit = (byte == 0) | (byte == 255)
        flag = ((z >> np.uint32(8 * t + 7)) & np.uint32(1)) == 1
        assert np.all(flag[hit])
    g = rng.integers(1, 255, (100000, 4), dtype=np.uint64)                           # no byte is 0 or 255: (almost) nothing may be flagged
    g = (g[:, 0] | g[:, 1] << np.uint64(8) | g[:, 2] << np.uint64(16) | g[:, 3] << np.uint64(24)).astype(np.uint32)
    y = (g ^ (g << np.uint32(1))) & np.uint32(0xFEFEFEFE)
    z = ~y & (y - np.uint32(0x01010101)) & np.uint32(0x80808080)
    assert (z != 0).mean() < 0.01


def _lane_window_max_offset(s, W):
    """Largest byte offset a lane of k_pyramid reads inside its slot: right tap of its last column relative to the dword-aligned start
    (the kernel's f32 expressions: xl = floor(fl(s * w)))."""
    w = np.arange(0, W, 4)
    x0 = np.floor(np.float32(s) * w.astype(np.float32)).astype(np.int64)
    x3 = np.floor(np.float32(s) * np.minimum(w + 3, W - 1).astype(np.float32)).astype(np.int64)
    return int((x3 + 1 - (x0 & ~3)).max())


def test_loads_per_row_cover_the_lane_window():
    """pyramid_loads_per_row (k_pyramid.hip): the 16-byte loads of a lane must cover byte offset floor(3 s) + 5 inclusive in the worst case (the round-3
    budget of floor(3 s) + 5 BYTES was one short: scaleFactor 1.25 with 7 levels, 1.3 with 6, 1.4 with 5 sampled past their slot).  The product now
    enumerates the level's lanes; this restates that enumeration and pins the counter-example of the round-3 review (s = 1.25^6, w = 44)."""
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jetson_slam_amd", "csrc", "k_pyramid.hip")).read()
    assert "max_off / 16 + 1" in src and "x3 + 1 - (x0 & ~3)" in src
    s = np.float32(1.0)
    for _ in range(6):
        s = np.float32(s * np.float32(1.25))
    s = np.float32(1.0) / (np.float32(1.0) / s)
    xl = [int(np.floor(s * np.float32(44 + t))) for t in range(4)]
    assert xl == [167, 171, 175, 179] and xl[3] + 1 - (xl[0] & ~3) == 16           # a second 16-byte load is needed
    assert _lane_window_max_offset(s, 197) // 16 + 1 == 2
    rng = np.random.default_rng(5)
    for sc in list(rng.uniform(1.0, 18.0, 300)) + [1.2 ** k for k in range(1, 12)] + [1.25 ** 6, 1.3 ** 5, 1.4 ** 4, 3.05 ** 2, 2.0, 4.0, 8.0]:
        for W in (37, 209, 1241, 4096):
            off = _lane_window_max_offset(sc, W)
            assert off <= int(np.float32(3.0) * np.float32(sc)) + 5 + 1            # +1: f32 rounding of s * w may cross an integer
            assert off // 16 + 1 <= 4 or sc > 18
    # every level of the three BASELINE pyramids still takes ONE load per lane and row (no change of the benchmarked kernels)
    for W0 in (752, 1241, 1280):
        sc = np.float32(1.0)
        for lvl in range(1, 8):
            sc = np.float32(sc * np.float32(1.2))
            s_k = np.float32(1.0) / (np.float32(1.0) / sc)
            Wl = int(np.round(W0 / sc))
            assert _lane_window_max_offset(s_k, Wl) <= 15, (W0, lvl)
