"""CPU checks of host-evaluable logic behind k_describe and k_stereo (no GPU):
  * the constant tables of k_describe (jetson_slam_amd/csrc/describe_tables.h, compiled here with g++): FP8 codes of the pattern
    against the OCP E4M3 definition, the LDS slot permutation, and the moment multipliers - the row-pair dot-product evaluation of
    the intensity centroid, replayed in numpy exactly as the kernel's lanes do it, against the brute-force sums over the disc
    (orb_FAST_orientation.cu:17-65) for every alignment of the staged patch;
  * the exact run of scan-line buckets k_stereo computes per level (k_stereo.hip, `below` / `above` walk): replayed in float32 for
    every row and a range of scales against the reference's row test floor(y - r) <= vL <= ceil(y + r) (orb_stereo_match.cu:119-140)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dt") / "describe_tables_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "jetson_slam_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "describe_tables_dump.cpp")])
    return json.loads(subprocess.check_output([exe]))


def _e4m3(byte):
    """OCP FP8 E4M3 (bias 7, no infinities), the format v_cvt_pk_f32_fp8 decodes on gfx950"""
    s, e, m = byte >> 7, (byte >> 3) & 15, byte & 7
    v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
    return -v if s else v


def test_pattern_fp8_codes_decode_to_the_pattern_and_slots_are_a_permutation(tables):
    q, slot, X, Y = tables["pattern_q"], tables["slot"], tables["x"], tables["y"]
    assert sorted(slot) == list(range(256))
    assert min(X + Y) >= -13 and max(X + Y) <= 13                     # E4M3 holds the integers up to 16 exactly
    for b in range(256):
        w = q[slot[b]]
        got = [_e4m3((w >> (8 * k)) & 0xFF) for k in range(4)]
        assert got == [X[2 * b], X[2 * b + 1], Y[2 * b], Y[2 * b + 1]], b
    # lane sl reads four 16-byte chunks at dwords sl*16 + 4*((c + sl//4) % 4): the 16 lanes of a keypoint hit 16 different groups of 4 banks
    for c in range(4):
        groups = {((sl * 16 + 4 * ((c + sl // 4) % 4)) % 64) // 4 for sl in range(16)}
        assert len(groups) == 16
    # and the step-major order: chunk c of lane sl holds steps 4c .. 4c+3 (descriptor bits it*16 + sl)
    for sl in range(16):
        for it in range(16):
            assert slot[it * 16 + sl] == sl * 16 + 4 * (((it >> 2) + (sl >> 2)) & 3) + (it & 3)


def test_moment_multipliers_reproduce_the_disc_sums_for_every_alignment(tables):
    umax = tables["umax"]
    assert umax == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    mom = np.array(tables["moment"], dtype=np.uint64).reshape(8, 16, 2)
    rng = np.random.default_rng(5)
    for trial in range(24):
        a = trial % 8                                                  # (x - 15) & 7: offset of the keypoint's column -15 inside the staged row
        rows = rng.integers(0, 256, (31, 48), dtype=np.int64)          # staged rows: 48 bytes from the 8-byte aligned column
        if trial >= 16:
            rows[:] = 255                                              # saturated patch: the largest sums
        # brute force (reference): m10 = sum u*I, m01 = sum v*I over |u| <= umax[|v|]
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                px = int(rows[15 + v, a + 15 + u])
                m10 += u * px
                m01 += v * px
        # the kernel's evaluation: lane v owns rows 15 + v and 15 - v; step d takes the dword of columns -15 + 4d .. -12 + 4d
        # (bytes a + 4d .. a + 4d + 3 of the staged row: two aligned dwords + alignbyte) into dot products with the table's multipliers
        g10 = g01 = 0
        for lane in range(16):
            p1 = p2 = n1 = n2 = s1 = s2 = 0
            for d in range(8):
                cu, inn = int(mom[d, lane, 0]), int(mom[d, lane, 1])
                w1 = rows[15 + lane, a + 4 * d:a + 4 * d + 4]
                w2 = rows[15 - lane, a + 4 * d:a + 4 * d + 4]
                mu = [(cu >> (8 * j)) & 0xFF for j in range(4)]
                mi = [(inn >> (8 * j)) & 0xFF for j in range(4)]
                du1, du2 = int(np.dot(w1, mu)), int(np.dot(w2, mu))
                if d < 4:
                    n1 += du1; n2 += du2
                else:
                    p1 += du1; p2 += du2
                s1 += int(np.dot(w1, mi)); s2 += int(np.dot(w2, mi))
                assert max(p1, p2, n1, n2, s1, s2) < 2 ** 32            # u32 accumulators of v_dot4_u32_u8
            g10 += (p1 - n1) + ((p2 - n2) if lane else 0)
            g01 += lane * (s1 - s2)
        assert (g10, g01) == (m10, m01), (trial, a)
    # the widest read of a lane: dwords at bytes (a & 4) + 4d and + 4 more, shifted by a & 3 -> last byte a + 4*7 + 3 + (4 - ...) stays inside the 48-byte row
    assert 7 + 4 * 7 + 3 < 48 and (7 & 4) + 4 * 8 + 3 < 48


def _f32(x):
    return np.float32(x)


def test_scan_line_bucket_runs_equal_the_references_row_test():
    """k_stereo.hip: lo = floor(vL - 2 - r), hi = ceil(vL + 2 + r); `below` = how many of lo .. lo+4 have ceil(y + r) < vL, `above` = how many
    of hi .. hi-4 have floor(y - r) > vL; the run is [lo + below, hi - above].  Must equal {y : floor(y - r) <= vL <= ceil(y + r)} with every
    sum rounded to float32 as on the device, for every row and scale (the monotonicity argument of DESIGN.md, checked by enumeration)."""
    scales = []
    for sf in (1.2, 1.1, 1.5, 2.0, 1.05, 1.25):
        s = np.float32(1.0)
        for lvl in range(12):
            scales.append(s)
            s = np.float32(s * np.float32(sf))
    scales = sorted({float(x) for x in scales if x < 19.0})          # (the library rejects level scales of 19 and more)
    ys = np.arange(-128, 8300, dtype=np.float32)
    checked = 0
    for sc in scales:
        r = _f32(2.0) * _f32(sc)
        lo_row = np.floor(ys - r)              # float32 arithmetic throughout (ys, r are float32)
        hi_row = np.ceil(ys + r)
        assert np.all(np.diff(lo_row) >= 0) and np.all(np.diff(hi_row) >= 0)      # non-decreasing in y
        for vL in list(range(0, 64)) + list(range(470, 490)) + list(range(1000, 1100, 7)) + list(range(2040, 2060)) + list(range(4090, 4100)) + [8191]:
            vLf = _f32(vL)
            lo_f = np.floor(vLf - _f32(2.0) - r)
            hi_f = np.ceil(vLf + _f32(2.0) + r)
            below = sum(1 for k in range(5) if np.ceil((lo_f + _f32(k)) + r) < vLf)
            above = sum(1 for k in range(5) if np.floor((hi_f - _f32(k)) - r) > vLf)
            ylo, yhi = int(lo_f) + below, int(hi_f) - above
            passing = ys[(lo_row <= vLf) & (vLf <= hi_row)].astype(np.int64)
            assert passing.size > 0 and passing[0] == ylo and passing[-1] == yhi and passing.size == yhi - ylo + 1, (sc, vL, ylo, yhi, passing[:3], passing[-3:])
            assert below < 5 and above < 5                             # the walk never runs out of its five candidates
            checked += 1
    assert checked > 5000


def test_bucket_candidate_search_finds_exactly_the_references_candidates():
    """The candidate search of k_compact + k_stereo restated in numpy on the ORACLE's keypoints of a stereo pair: every right keypoint listed
    once under (level, level-0 row), the run of buckets per level from the walk above, the disparity-window test per candidate - against the
    reference's definition (row table over [floor(y - r), ceil(y + r)], octave within +-1, uR in [uL - maxD, uL]; orb_stereo_match.cu:119-184)
    evaluated by brute force, and against the oracle's own count of candidate pairs."""
    from jetson_slam_amd.synth import synth_stereo_pair
    from oracle import pyoracle as po
    H, W, L, tile, th, fx, bf = 376, 620, 8, 25, 60, 718.856, 386.1448
    l, r = synth_stereo_pair(77, H, W)
    mk = lambda: po.OracleExtractor(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
    ol, orr = mk(), mk()
    ol.extract(l); orr.extract(r)
    mb = bf / fx
    _, _, ost = po.stereo_match(ol, orr, mb, bf)
    kl, kr = ol.keypoints().reshape(6, -1), orr.keypoints().reshape(6, -1)
    scales = ol.scales()
    xl, yl, lvl_l = kl[0], kl[1], kl[4]
    xr, yr, lvl_r = kr[0], kr[1], kr[4]
    assert len(xl) > 300 and len(xr) > 300 and yr.max() < H and xr.max() < 32768
    minD, maxD = np.float32(0.0), np.float32(np.float32(bf) / np.float32(mb))      # Frame.cpp: maxD = mbf / mb
    # --- buckets as k_compact builds them: (level, row) -> list of right keypoint indices
    buckets = {}
    for j in range(len(xr)):
        buckets.setdefault((int(lvl_r[j]), min(int(yr[j]), H - 1)), []).append(j)
    total_ref = total_bkt = 0
    for i in range(len(xl)):
        uL, vL, lv = np.float32(xl[i]), np.float32(yl[i]), int(lvl_l[i])
        vLi = int(vL)
        minU, maxU = uL - maxD, uL - minD
        # reference definition, brute force over all right keypoints
        ref = set()
        for j in range(len(xr)):
            rr = np.float32(2.0) * scales[lvl_r[j]]
            kpY = np.float32(yr[j])
            if int(np.floor(kpY - rr)) <= vLi <= int(np.ceil(kpY + rr)) and abs(int(lvl_r[j]) - lv) <= 1:
                uR = np.float32(xr[j])
                if minU <= uR <= maxU and not maxU < 0:
                    ref.add(j)
        # the kernel's search
        got = set()
        if not maxU < 0:
            for lr in (lv - 1, lv, lv + 1):
                if lr < 0 or lr >= L:
                    continue
                rr = np.float32(2.0) * scales[lr]
                vLf = np.float32(vLi)
                lo_f, hi_f = np.floor(vLf - np.float32(2.0) - rr), np.ceil(vLf + np.float32(2.0) + rr)
                below = sum(1 for k in range(5) if np.ceil((lo_f + np.float32(k)) + rr) < vLf)
                above = sum(1 for k in range(5) if np.floor((hi_f - np.float32(k)) - rr) > vLf)
                for y in range(max(int(lo_f) + below, 0), min(int(hi_f) - above, H - 1) + 1):
                    for j in buckets.get((lr, y), ()):
                        if minU <= np.float32(xr[j]) <= maxU:
                            got.add(j)
        assert got == ref, (i, sorted(got ^ ref)[:5])
        total_ref += len(ref); total_bkt += len(got)
    assert total_bkt == total_ref == ost["n_candidate_pairs"]


def test_detect_counted_loop_visits_exactly_the_non_border_steps():
    """k_detect.hip: the early-reject steps of wave w are rbase = w*rps, + NW*rps, ... < score_rows, minus those whose rows all lie in the
    image's 20-pixel border (round-3 first half: `continue` inside the loop).  The counted loop of the second half starts at
    rb_first = first step >= BORDER - (y0-1) - (rps-1) on that lattice and ends below rb_end = min(score_rows, H - BORDER - (y0-1)).
    Same set of steps for every band position, image height, region height, wave and row pairing."""
    BORDER, NW = 20, 4
    checked = 0
    for rps in (1, 2):
        step_rows = NW * rps
        sh = {4: 2, 8: 3}[step_rows]
        for H in (60, 97, 240, 376, 480):
            for score_rows in (10, 17, 32, 62, 122):
                for y0 in range(0, H, 7):
                    for wave in range(NW):
                        old = [rb for rb in range(wave * rps, score_rows, step_rows)
                               if not (y0 - 1 + rb + rps - 1 < BORDER or y0 - 1 + rb >= H - BORDER)]
                        rb_first = wave * rps
                        rb_min = BORDER - (y0 - 1) - (rps - 1)
                        if rb_first < rb_min:
                            rb_first += ((rb_min - rb_first + step_rows - 1) >> sh) << sh
                        rb_end = min(score_rows, H - BORDER - (y0 - 1))
                        new = list(range(rb_first, rb_end, step_rows))
                        assert new == old, (rps, H, score_rows, y0, wave)
                        checked += 1
    assert checked > 3000


def test_three_dimensional_grid_enumerates_the_same_image_block_pairs():
    """jsorb_device.h xcd_grid / xcd_map: grid (8, nb, ceil(n/8)) with b = z*8 + x, blk = y for batches of 8 and more images (workgroups of one
    image on one XCD when dispatched x-fastest), (nb, n) below; the linear form (nb > 65535) divides.  Every (image, block) pair exactly once,
    and the x-fastest linear order of the 3-D grid is the linear form's order."""
    def grid3(nb, n):
        return (8, nb, (n + 7) // 8) if n >= 8 else (nb, n, 1)

    def map3(x, y, z, nb, n):
        return (z * 8 + x, y) if n >= 8 else (y, x)

    def map_linear(lin, nb, n):
        if n >= 8:
            q = lin >> 3
            return ((q // nb) * 8 + (lin & 7), q - (q // nb) * nb)
        return (lin // nb, lin - (lin // nb) * nb)

    for nb in (1, 3, 57, 220):
        for n in (1, 2, 7, 8, 9, 16, 33):
            gx, gy, gz = grid3(nb, n)
            seen, lin = [], 0
            for z in range(gz):
                for y in range(gy):
                    for x in range(gx):
                        b, blk = map3(x, y, z, nb, n)
                        assert (b, blk) == map_linear(lin, nb, n)
                        lin += 1
                        if b < n:
                            seen.append((b, blk))
            assert sorted(seen) == [(b, k) for b in range(n) for k in range(nb)]


def test_six_bit_early_rejects_accept_a_superset_for_every_threshold():
    """k_detect.hip detect_swar6_threshold (round 4): with q(x) = x >> 2 and t4 = (th + 1) >> 2, `p > v + th` implies bit 7 of q(p) + (128 - t4 - q(v))
    and `p < v - th` implies bit 7 of (128 - t4 + q(v)) - q(p); both sums stay inside a byte, so four pixels share a dword without carries.  The host
    checks this exhaustively before it selects the 6-bit kernel; restated here for every threshold, together with the monotonicity the compass test
    needs (more flags never turn an accepted pixel into a rejected one)."""
    src = open(os.path.join(ROOT, "jetson_slam_amd", "csrc", "k_detect.hip")).read()
    assert "(p >> 2) + (cb - (v >> 2))" in src and "(cb + (v >> 2)) - (p >> 2)" in src and "t4 < 2" in src
    v = np.arange(256)[:, None]
    p = np.arange(256)[None, :]
    for th in range(0, 300):
        thc = min(th, 256)
        t4 = (thc + 1) >> 2
        cb = 128 - t4
        y = (p >> 2) + (cb - (v >> 2))
        z = (cb + (v >> 2)) - (p >> 2)
        assert y.min() >= 0 and y.max() <= 255 and z.min() >= 0 and z.max() <= 255 and cb - 63 >= 0 and cb + 63 <= 255
        assert np.all(((y & 0x80) != 0)[p > v + th]) and np.all(((z & 0x80) != 0)[p < v - th])
    # compass combination: (B4|B12)&(B0|B8) | (D4|D12)&(D0|D8) is monotone in its eight flags
    for m in range(256):
        f = [(m >> k) & 1 for k in range(8)]
        acc = ((f[0] | f[1]) & (f[2] | f[3])) | ((f[4] | f[5]) & (f[6] | f[7]))
        for k in range(8):
            if not f[k]:
                f2 = list(f); f2[k] = 1
                assert (((f2[0] | f2[1]) & (f2[2] | f2[3])) | ((f2[4] | f2[5]) & (f2[6] | f2[7]))) >= acc
