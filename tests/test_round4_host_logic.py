"""CPU restatements of the round-4 host logic and of the index arithmetic the round-4 kernels rely on (no GPU, no oracle):

* k_describe: which patch row a quad of lanes stages in which step - every row of the 31-row disc is written, and no two dwords of a 16-lane
  store group share an LDS bank (bank model of MI355X_MICROARCH.md: ds_write_b64 is served in four groups of 16 consecutive lanes, bank = dword mod 32);
  the round-3 assignment (rows q + 4k) is shown to collide two-way by the same model.
* k_stereo: the flat (keypoint, row) task list - rank = (t * 745) >> 13 is t // 11 over the whole range, 16-bit packed sums cannot carry, every
  (rank, row) pair is visited exactly once; the passes-per-wave rule.
* jsorb_create: the LDS request of k_detect that leaves exactly one k_describe workgroup per CU.
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "jetson_slam_amd", "csrc")


def _src(name):
    return open(os.path.join(CSRC, name)).read()


def _define(src, name):
    m = re.search(r"#define\s+%s\s+\(?([0-9]+)" % name, src)
    assert m, name
    return int(m.group(1))


def _define_row40(src, name):
    return _define(src, name)


def _describe_rows(bq, layout):
    """rows of the un-blurred patch a quad stages in steps 0..7 (k_describe.hip: row0, row6, row7)"""
    if layout == "round3":                            # rows q + 4k, last step min(28 + q, 30)
        return [bq + 4 * k for k in range(7)] + [min(28 + bq, 30)]
    if layout == "round4":                            # quads 2 and 3 one step ahead (56-byte rows)
        row0 = bq + (4 if bq >= 2 else 0)
        return [row0 + 4 * k for k in range(6)] + [min(row0 + 24, 30), bq if bq >= 2 else 28 + bq]
    return [bq + 4 * k for k in range(8)]             # round 6, 40-byte rows: rows 4k + q, step 7 = rows 28 .. 31


def _store_cycles(layout, stride, patch_bytes):
    """LDS cycles of the ds_write_b64 halves of one wave (4 keypoints x 4 quads x 3 storing lanes), one cycle per group when conflict-free; with 40-byte
    rows the third unit stores only its first half"""
    total = 0
    for k in range(8):
        for half in range(2):
            for grp in range(4):                      # a 16-lane group = one keypoint
                banks = {}
                for sl in range(16):
                    bq, bu = sl >> 2, sl & 3
                    if bu == 3 or (layout == "round6" and bu == 2 and half == 1):
                        continue                      # the fourth lane of a quad repeats unit 2 and stores nothing
                    a = grp * patch_bytes + _describe_rows(bq, layout)[k] * stride + 16 * bu + 8 * half
                    for dw in (a // 4, a // 4 + 1):
                        banks.setdefault(dw % 32, set()).add(dw)
                total += max(len(v) for v in banks.values())
    return total


def test_describe_patch_staging_covers_every_row_and_its_stores_do_not_share_banks():
    src = _src("k_describe.hip")
    stride = _define_row40(src, "ORI_LDS_STRIDE")
    patch_bytes = 37 * _define_row40(src, "BLR_Q") * 8
    assert stride == 40 and patch_bytes == 1480
    assert "const int row0 = bq;" in src and "const int row6 = row0 + 24, row7 = row0 + 28;" in src
    rows = sorted(r for bq in range(4) for r in _describe_rows(bq, "round6"))
    assert rows == list(range(32))                                          # rows 0 .. 30 are the disc, row 31 is staged too and lies inside the region
    assert max(rows) * stride + 40 <= patch_bytes
    # a window of 40 bytes from the 4-byte aligned column below the first pixel holds the 31 resp. 37 pixels of a patch row
    assert 3 + 31 <= 40 and 3 + 37 <= 40
    # 10-dword rows: the four consecutive rows of a store group never share a bank (no staggering needed); the 48 / 56-byte layouts of rounds 3 / 4 for comparison
    new = _store_cycles("round6", stride, patch_bytes)
    assert new == 8 * 2 * 4, new                                           # one cycle per 16-lane group: conflict-free
    assert _store_cycles("round3", 56, 1776) >= 2 * 14 * 4 and _store_cycles("round4", 56, 1776) <= 16 * 4 + 2 * 4
    # the 16 rows y - 15 + v .. a keypoint's lanes read at once for the moments start in 16 different banks
    assert len({(r * stride // 4) % 32 for r in range(16)}) == 16
    # workgroups per CU: 16 keypoints x 1480 B + pattern + level table + moment table = 25 984 B = 21 granules -> SIX per CU (five with 1776-byte regions)
    kpwg = _define(src, "KPW") * _define(src, "WPW")
    static_lds = kpwg * patch_bytes + 256 * 4 + 16 * 16 + 8 * 16 * 2 * 4
    assert static_lds == 25984 and (160 * 1024) // (-(-static_lds // 1280) * 1280) == 6
    assert _define(src, "DESC_MIN_WAVES") == 6                              # ... and the register allocation must allow the six waves per SIMD


def test_stereo_flat_task_list_index_arithmetic():
    src = _src("k_stereo.hip")
    assert "(tt * 745) >> 13" in src and "tt - 11 * ridx" in src
    max_kp = _define(src, "SKPW") * _define(src, "ST_MAX_PASS")
    t = np.arange(0, 2700)
    assert np.array_equal((t * 745) >> 13, t // 11)
    assert 11 * max_kp <= 2700
    for n_ref in (0, 1, 5, max_kp):
        seen = set()
        for t0 in range(0, 11 * n_ref, 64):
            for lane in range(64):
                tt = t0 + lane
                if tt < 11 * n_ref:
                    rank = (tt * 745) >> 13
                    seen.add((rank, tt - 11 * rank))
        assert seen == {(k, r) for k in range(n_ref) for r in range(11)}
    # two 16-bit sums per dword: a window's L1 sum is < 2^16, so the packed LDS additions never carry from one field into the other
    assert 11 * 11 * 510 < 1 << 16


def test_stereo_passes_per_wave_rule():
    """launch_stereo: 8 passes if that leaves >= 24 k waves, else 6 / 4 / 2 with >= 12 k, a single pair 1 (k_stereo.hip)"""
    src = _src("k_stereo.hip")
    assert "waves(8) >= 24576 ? 8 : waves(6) >= 12288 ? 6 : waves(4) >= 12288 ? 4 : waves(2) >= 12288 ? 2 : 1" in src

    def npass(n_pairs, T):
        if n_pairs == 1:
            return 1
        waves = lambda np_: n_pairs * ((T + 4 * np_ - 1) // (4 * np_))
        return 8 if waves(8) >= 24576 else 6 if waves(6) >= 12288 else 4 if waves(4) >= 12288 else 2 if waves(2) >= 12288 else 1
    assert npass(128, 3466) == 6 and npass(64, 6756) == 6 and npass(64, 21053) == 8 and npass(1, 3466) == 1 and npass(2, 3466) == 1
    assert npass(16, 3466) == 1 and npass(32, 3466) == 2 and npass(64, 3466) == 4


def test_detect_lds_request_leaves_one_describe_workgroup_per_cu():
    """jsorb_create: request = (160 KB - k_describe's LDS rounded up to 1280-byte granules) / 4, rounded down to granules"""
    src = _src("jsorb_api.hip")
    assert "(cu_lds - desc) / 4 / gran * gran" in src and "gran = 1280" in src
    d = _src("k_describe.hip")
    kpwg = _define(d, "KPW") * _define(d, "WPW")
    static_lds = kpwg * 37 * _define_row40(d, "BLR_Q") * 8 + 256 * 4 + 16 * 16 + 8 * 16 * 2 * 4      # patches, pattern, level table (JSORB_MAX_LEVELS int4), moment table
    gran, cu = 1280, 160 * 1024
    desc = (static_lds + gran - 1) // gran * gran
    want = (cu - desc) // 4 // gran * gran
    assert static_lds == 25984 and want == 33280
    assert 4 * want + desc <= cu < 4 * (want + gran) + desc                 # four k_detect workgroups + one k_describe workgroup fit, a granule more does not
    assert 5 * want > cu                                                   # and a fifth k_detect workgroup does not fit
