"""k_blur's certified fast path (jetson_slam_amd/csrc/k_blur.hip): a separable 7+7-FMA evaluation A of the 7x7 blur decides floor(C) of the
reference's 49-FMA chain C whenever A is farther than BLUR_BAND = 2^-8 from an integer (floor(256 A) mod 256 is neither 0 nor 255, read
off the mantissa of A + 49152 rounded down); the other pixels are recomputed exactly.  This test re-derives the rigorous bound
|A - C| <= bound from the constants IN THE KERNEL SOURCE with exact rational arithmetic and requires BLUR_BAND to cover it with margin,
and checks the claim empirically on adversarial windows (the bit-exact -m gpu plane comparisons are what finally guard the kernel)."""
import os
import re
from fractions import Fraction

import numpy as np

from oracle import host_restatement as hr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = Fraction(1, 2 ** 24)


def _kernel_constants():
    src = open(os.path.join(ROOT, "jetson_slam_amd", "csrc", "k_blur_body.h")).read()
    delta = float(re.search(r"#define BLUR_BAND ([0-9.e+-]+)f", src).group(1))
    assert delta == 2.0 ** -8 and "const float magic = 49152.0f;" in src         # the band IS the ulp of the magic number
    assert src.index("// horizontal stage") < src.index("// vertical stage")    # stage order of the kernel (round 4): horizontal sums first, vertical sum of those
    body = src[src.index("__constant__ float c_sep_v[4]"):]
    v = [float(t) for t in re.findall(r"([0-9.]+)f", body[:body.index(";")])]
    hb = body[body.index("c_sep_h[4]"):]
    hb = hb[:hb.index(";")]
    h = [eval(t, {"__builtins__": {}}, {}) for t in re.findall(r"\(float\)\(([^)]*)\)", hb)]
    return delta, np.array(v, np.float32), np.array(h, np.float32)


def _gamma(n):
    return n * U / (1 - n * U)


def test_delta_covers_the_rigorous_error_bound():
    delta, v4, h4 = _kernel_constants()
    w = hr.CtorTables._gauss().reshape(7, 7)                              # the reference's f32 weights (glibc expf, f32 normalisation)
    gv = [Fraction(float(v4[abs(j)])) for j in range(-3, 4)]
    gh = [Fraction(float(h4[abs(k)])) for k in range(-3, 4)]
    wq = [[Fraction(float(w[j, k])) for k in range(7)] for j in range(7)]
    sum_w = sum(sum(r) for r in wq)
    # chain of 49 FMAs: every term passes through at most 49 roundings
    bound_c = _gamma(49) * 255 * sum_w
    # separable: 7-FMA horizontal stage on values <= 255 * sum(gh), then 7-FMA vertical stage on its results
    hmax = 255 * sum(gh)
    bound_h = _gamma(7) * hmax
    bound_a = _gamma(7) * sum(gv) * (hmax + bound_h) + sum(gv) * bound_h
    model = 255 * sum(abs(gv[j] * gh[k] - wq[j][k]) for j in range(7) for k in range(7))
    total = bound_c + bound_a + model                                     # (A + 49152 rounded down is floor(256 A) / 256 exactly: no further error)
    assert float(total) < 1.0e-3, float(total)
    assert delta >= 3.5 * float(total)                                    # the band is several times a bound that is itself worst-case
    assert delta < 0.01                                                   # ... without listing more than ~2 % of natural pixels


def test_certificate_decides_correctly_on_adversarial_windows():
    delta, v4, h4 = _kernel_constants()
    w = hr.CtorTables._gauss().reshape(7, 7)
    gv = np.array([v4[abs(j)] for j in range(-3, 4)], np.float32)
    gh = np.array([h4[abs(k)] for k in range(-3, 4)], np.float32)
    rng = np.random.default_rng(11)
    n = 200000
    wins = [rng.integers(0, 256, (n, 7, 7), dtype=np.uint8),
            np.repeat(np.arange(256, dtype=np.uint8), 8).reshape(-1, 1, 1).repeat(7, 1).repeat(7, 2),        # exactly flat windows: C within 1e-4 of an integer
            rng.integers(250, 256, (n, 7, 7)).astype(np.uint8),
            np.clip(rng.integers(0, 256, (n, 1, 1)) + rng.integers(-1, 2, (n, 7, 7)), 0, 255).astype(np.uint8),
            np.zeros((16, 7, 7), np.uint8)]
    decided = 0
    for P in wins:
        acc = np.zeros(P.shape[0], np.float32)
        for j in range(7):
            for k in range(7):                                            # f32 fma via f64 (exact product, one extra rounding far below the margins)
                acc = (np.float64(w[j, k]) * P[:, j, k] + acc.astype(np.float64)).astype(np.float32)
        Hs = np.zeros((P.shape[0], 7), np.float32)                         # horizontal sums of the 7 window rows (left to right)
        for k in range(7):
            Hs = (np.float64(gh[k]) * P[:, :, k] + Hs.astype(np.float64)).astype(np.float32)
        A = np.zeros(P.shape[0], np.float32)
        for j in (0, 1, 2, 3, 4, 5, 6):                                    # vertical sum of those, oldest row first
            A = (np.float64(gv[j]) * Hs[:, j] + A.astype(np.float64)).astype(np.float32)
        assert np.abs(A.astype(np.float64) - acc.astype(np.float64)).max() < 0.3 * delta
        q = np.floor(A.astype(np.float64) * 256.0).astype(np.int64)       # mantissa of A + 49152 rounded down
        frac, ipart = q & 0xFF, q >> 8
        sure = (frac != 0) & (frac != 255)
        assert np.array_equal(ipart[sure], np.floor(acc[sure]).astype(np.int64))       # the certificate never lies
        decided += int(sure.sum())
    assert decided > 0.95 * 3 * n                                         # and it decides almost every natural pixel
