"""ctypes binding of the CPU oracle (oracle/libjsorb_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg - never from the product package (jetson_slam_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcParams(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("n_levels", C.c_int), ("scale_factor", C.c_float),
                ("fast_n_min", C.c_int), ("fast_n_max", C.c_int), ("th_fast_min", C.c_int), ("th_fast_max", C.c_int),
                ("tile_h", C.c_int), ("tile_w", C.c_int), ("fixed_multi_scale_tile_size", C.c_int),
                ("apply_nms_ms", C.c_int), ("nms_ms_mode_gpu", C.c_int)]


class OrcStereoStats(C.Structure):
    _fields_ = [("n_left", C.c_int), ("n_right", C.c_int), ("n_candidate_pairs", C.c_int), ("n_corr_match", C.c_int),
                ("n_depth", C.c_int), ("n_final", C.c_int), ("n_row_oob", C.c_int)]


def build(native=False):
    target = "native" if native else "all"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])


def _load(native=False):
    name = "libjsorb_oracle_native.so" if native else "libjsorb_oracle.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build(native)
    lib = C.CDLL(path)
    P = C.c_void_p
    sig = {
        "orc_create": (P, [C.POINTER(OrcParams), C.c_void_p]),
        "orc_destroy": (None, [P]),
        "orc_extract": (C.c_int, [P, C.c_void_p, C.c_int]),
        "orc_n_keypoints": (C.c_int, [P]),
        "orc_out_keypoints": (C.POINTER(C.c_int32), [P]),
        "orc_out_descriptors": (C.POINTER(C.c_uint8), [P]),
        "orc_n_levels": (C.c_int, [P]),
        "orc_total_tiles": (C.c_int, [P]),
        "orc_fast_lut": (C.POINTER(C.c_uint8), [P]),
        "orc_umax": (C.POINTER(C.c_int32), [P]),
        "orc_gauss_weights": (C.POINTER(C.c_float), [P]),
        "orc_tile_x": (C.POINTER(C.c_int32), [P]),
        "orc_tile_y": (C.POINTER(C.c_int32), [P]),
        "orc_tile_score": (C.POINTER(C.c_int32), [P]),
        "orc_stereo_best_right": (C.POINTER(C.c_int32), [P]),
        "orc_stereo_best_dist": (C.POINTER(C.c_int32), [P]),
        "orc_stereo_l1": (C.POINTER(C.c_int32), [P]),
        "orc_atan2f": (C.c_float, [C.c_float, C.c_float]),
        "orc_cosf": (C.c_float, [C.c_float]),
        "orc_sinf": (C.c_float, [C.c_float]),
        "orc_bilinear_px": (C.c_uint8, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int]),
        "orc_gauss_px": (C.c_uint8, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
        "orc_fast_score_px": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]),
        "orc_hamming256": (C.c_int, [C.c_void_p, C.c_void_p]),
        "orc_desc_offset": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]),
        "orc_nms_tiles_plane": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_nms_ms_gpu_candidates": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "orc_orientation_px": (C.c_float, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]),
        "orc_descriptor_px": (None, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
        "orc_pack_level": (None, [C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
        "orc_l1_sums": (None, [C.c_int] + [C.c_void_p] * 8),
        "orc_project_points": (None, [C.c_int] + [C.c_void_p] * 5 + [C.c_float] * 8 + [C.c_void_p] * 4),
        "orc_hamming_pairs": (None, [C.c_int] + [C.c_void_p] * 5),
        "orc_logf": (C.c_float, [C.c_float]),
        "orc_unpack_keypoints": (None, [C.c_int, C.c_void_p, C.c_void_p]),
        "orc_assign_features_to_grid": (C.c_int, [C.c_int, C.c_void_p] + [C.c_float] * 4 + [C.c_int] * 2 + [C.c_void_p] * 2),
        "orc_is_in_frustum": (None, [C.c_int] + [C.c_void_p] * 12 + [C.c_float] * 4 + [C.c_int] * 5 + [C.c_float] * 2 + [C.c_void_p] * 6),
        "orc_bench_pairs": (C.c_long, [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_double,
                                      C.c_int, C.POINTER(C.c_double)]),
        "orc_pairs_digest": (C.c_int, [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
        "orc_stereo_match": (C.c_int, [P, P, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(OrcStereoStats)]),
    }
    for lvl_fn, rt in [("orc_level_height", C.c_int), ("orc_level_width", C.c_int), ("orc_level_scale", C.c_float),
                       ("orc_level_inv_scale", C.c_float), ("orc_tile_h", C.c_int), ("orc_tile_w", C.c_int),
                       ("orc_n_tile_h", C.c_int), ("orc_n_tile_w", C.c_int), ("orc_level_offset", C.c_int),
                       ("orc_level_n_keypoints", C.c_int),
                       ("orc_level_image", C.POINTER(C.c_uint8)), ("orc_level_blurred", C.POINTER(C.c_uint8)),
                       ("orc_level_score", C.POINTER(C.c_int32)), ("orc_level_mask", C.POINTER(C.c_uint8)), ("orc_kp_x", C.POINTER(C.c_int32)),
                       ("orc_kp_y", C.POINTER(C.c_int32)), ("orc_kp_score", C.POINTER(C.c_int32)),
                       ("orc_kp_angle", C.POINTER(C.c_float))]:
        sig[lvl_fn] = (rt, [P, C.c_int])
    for name, (rt, at) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = rt, at
    return lib


_LIBS = {}


def lib(native=False):
    if native not in _LIBS:
        _LIBS[native] = _load(native)
    return _LIBS[native]


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()


def make_params(height, width, n_levels=8, scale_factor=1.2, fast_n_min=9, fast_n_max=14, th_fast_min=7,
                th_fast_max=20, tile_h=30, tile_w=30, fixed_tile=False, apply_nms_ms=False, nms_ms_mode_gpu=False):
    return OrcParams(height, width, n_levels, scale_factor, fast_n_min, fast_n_max, th_fast_min, th_fast_max,
                     tile_h, tile_w, int(fixed_tile), int(apply_nms_ms), int(nms_ms_mode_gpu))


class OracleExtractor:
    """CPU restatement of ORB_GPU (orb_gpu.cpp) - checker only."""

    def __init__(self, native=False, mask=None, **kw):
        self.l = lib(native)
        self.params = make_params(**kw)
        self._mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.h = self.l.orc_create(C.byref(self.params), None if mask is None else self._mask.ctypes.data)
        if not self.h:
            raise ValueError("orc_create rejected the parameters")
        self.n_levels = self.l.orc_n_levels(self.h)
        self.T = self.l.orc_total_tiles(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.orc_destroy(self.h)
            self.h = None

    # geometry
    def level_dims(self):
        return [(self.l.orc_level_height(self.h, i), self.l.orc_level_width(self.h, i)) for i in range(self.n_levels)]

    def tile_dims(self):
        return [(self.l.orc_tile_h(self.h, i), self.l.orc_tile_w(self.h, i)) for i in range(self.n_levels)]

    def tile_grid(self):
        return [(self.l.orc_n_tile_h(self.h, i), self.l.orc_n_tile_w(self.h, i)) for i in range(self.n_levels)]

    def level_offsets(self):
        return [self.l.orc_level_offset(self.h, i) for i in range(self.n_levels)]

    def scales(self):
        return np.array([self.l.orc_level_scale(self.h, i) for i in range(self.n_levels)], np.float32)

    def inv_scales(self):
        return np.array([self.l.orc_level_inv_scale(self.h, i) for i in range(self.n_levels)], np.float32)

    def lut(self):
        return _arr(self.l.orc_fast_lut(self.h), 65536, np.uint8)

    def umax(self):
        return _arr(self.l.orc_umax(self.h), 16, np.int32)

    def gauss_weights(self):
        return _arr(self.l.orc_gauss_weights(self.h), 49, np.float32)

    # pipeline
    def extract(self, image):
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.params.height, self.params.width)
        self._img = image
        return self.l.orc_extract(self.h, image.ctypes.data, image.strides[0])

    @property
    def n(self):
        return self.l.orc_n_keypoints(self.h)

    def keypoints(self):
        """6N int32 SoA: x, y, score, angle(deg, f32 bits), octave, size."""
        return _arr(self.l.orc_out_keypoints(self.h), 6 * self.n, np.int32)

    def descriptors(self):
        return _arr(self.l.orc_out_descriptors(self.h), 32 * self.n, np.uint8).reshape(-1, 32)

    def level_image(self, i):
        h, w = self.level_dims()[i]
        return _arr(self.l.orc_level_image(self.h, i), h * w, np.uint8).reshape(h, w)

    def level_blurred(self, i):
        h, w = self.level_dims()[i]
        return _arr(self.l.orc_level_blurred(self.h, i), h * w, np.uint8).reshape(h, w)

    def level_score(self, i):
        h, w = self.level_dims()[i]
        return _arr(self.l.orc_level_score(self.h, i), h * w, np.int32).reshape(h, w)

    def level_mask(self, i):
        h, w = self.level_dims()[i]
        return _arr(self.l.orc_level_mask(self.h, i), h * w, np.uint8).reshape(h, w)

    def tiles(self):
        return (_arr(self.l.orc_tile_x(self.h), self.T, np.int32), _arr(self.l.orc_tile_y(self.h), self.T, np.int32),
                _arr(self.l.orc_tile_score(self.h), self.T, np.int32))

    def level_keypoints(self, i):
        n = self.l.orc_level_n_keypoints(self.h, i)
        return (_arr(self.l.orc_kp_x(self.h, i), n, np.int32), _arr(self.l.orc_kp_y(self.h, i), n, np.int32),
                _arr(self.l.orc_kp_score(self.h, i), n, np.int32), _arr(self.l.orc_kp_angle(self.h, i), n, np.float32))


def stereo_match(left, right, mb, mbf, th_high=100, th_low=50):
    """ORB_GPU::ORB_compute_stereo_match on the last extract of two OracleExtractor objects."""
    n = left.n
    u = np.full(max(n, 1), -1, np.float32)
    d = np.full(max(n, 1), -1, np.float32)
    st = OrcStereoStats()
    rc = left.l.orc_stereo_match(left.h, right.h, mb, mbf, th_high, th_low, u.ctypes.data, d.ctypes.data, C.byref(st))
    assert rc == 0
    stats = {k: getattr(st, k) for k, _ in OrcStereoStats._fields_}
    stats["best_right"] = _arr(left.l.orc_stereo_best_right(left.h), n, np.int32)
    stats["best_dist"] = _arr(left.l.orc_stereo_best_dist(left.h), n, np.int32)
    stats["l1"] = _arr(left.l.orc_stereo_l1(left.h), n, np.int32)
    return u[:n], d[:n], stats


def pack_level(x, y, score, angle, octave, scale, n_total=None, kp_offset=0, out=None, native=False):
    """K11 for one level (orc_pack_level): returns / fills the 6 x n_total int32 SoA."""
    x, y, score = (np.ascontiguousarray(a, np.int32) for a in (x, y, score))
    angle = np.ascontiguousarray(angle, np.float32)
    n = len(x)
    n_total = n if n_total is None else n_total
    if out is None:
        out = np.zeros(6 * n_total, np.int32)
    lib(native).orc_pack_level(n, octave, float(scale), x.ctypes.data, y.ctypes.data, score.ctypes.data, angle.ctypes.data, n_total, kp_offset,
                               out.ctypes.data)
    return out


def l1_sums(levels_left, levels_right, x_left, x_right, y, octave, native=False):
    """K13 + cublasSgemv (orc_l1_sums): [m, 11] float32 window sums; levels_* are lists of contiguous uint8 planes (pitch = width)."""
    levels_left = [np.ascontiguousarray(a, np.uint8) for a in levels_left]
    levels_right = [np.ascontiguousarray(a, np.uint8) for a in levels_right]
    x_left, x_right, y, octave = (np.ascontiguousarray(a, np.int32) for a in (x_left, x_right, y, octave))
    m = len(x_left)
    out = np.zeros((m, 11), np.float32)
    PA = C.c_void_p * len(levels_left)
    pl, pr = PA(*[a.ctypes.data for a in levels_left]), PA(*[a.ctypes.data for a in levels_right])
    widths = np.array([a.shape[1] for a in levels_left], np.int32)
    if m:
        lib(native).orc_l1_sums(m, x_left.ctypes.data, x_right.ctypes.data, y.ctypes.data, octave.ctypes.data, C.cast(pl, C.c_void_p), C.cast(pr, C.c_void_p),
                                widths.ctypes.data, out.ctypes.data)
    return out


def bench_pairs(lefts, rights, mb, mbf, seconds, n_threads, native=True, **kw):
    """cpu_baseline leg of bench.py: OpenMP over independent pairs, returns (pairs_done, elapsed_s)."""
    l = lib(native)
    p = make_params(**kw)
    lefts = np.ascontiguousarray(lefts, np.uint8)
    rights = np.ascontiguousarray(rights, np.uint8)
    el = C.c_double()
    n = l.orc_bench_pairs(C.byref(p), lefts.ctypes.data, rights.ctypes.data, lefts.shape[0], mb, mbf, seconds, n_threads, C.byref(el))
    return n, el.value


_DIGEST_M = np.uint64(0x9E3779B97F4A7C15)


def digest_arrays(arrays):
    """the checksum of orc_pairs_digest over the concatenated bytes of `arrays` (numpy, wrapping uint64 arithmetic)"""
    b = np.concatenate([np.ascontiguousarray(a).view(np.uint8).reshape(-1) for a in arrays]) if arrays else np.zeros(0, np.uint8)
    with np.errstate(over="ignore"):
        w = (np.arange(1, b.size + 1, dtype=np.uint64)) * _DIGEST_M
        return int(((b.astype(np.uint64) + np.uint64(1)) * w).sum(dtype=np.uint64))


def pairs_digest(lefts, rights, mb, mbf, n_threads, native=False, **kw):
    """(digest[n] uint64, counts[n, 3] int32) of extract(L) + extract(R) + stereo for every pair, OpenMP over pairs"""
    l = lib(native)
    p = make_params(**kw)
    lefts = np.ascontiguousarray(lefts, np.uint8)
    rights = np.ascontiguousarray(rights, np.uint8)
    n = lefts.shape[0]
    dg = np.zeros(n, np.uint64)
    cnt = np.zeros((n, 3), np.int32)
    rc = l.orc_pairs_digest(C.byref(p), lefts.ctypes.data, rights.ctypes.data, n, mb, mbf, n_threads, dg.ctypes.data, cnt.ctypes.data)
    assert rc == 0
    return dg, cnt


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def unpack_keypoints(soa, native=False):
    """Frame.cpp:119-196: keypoint SoA (6N int32) -> structured array with the memory layout of cv::KeyPoint."""
    soa = np.ascontiguousarray(soa, np.int32)
    n = soa.size // 6
    out = np.zeros(n, KEYPOINT_DTYPE)
    lib(native).orc_unpack_keypoints(n, soa.ctypes.data, out.ctypes.data)
    return out


def assign_features_to_grid(soa, min_x, min_y, inv_w, inv_h, cols=64, rows=48, native=False):
    """Frame::AssignFeaturesToGrid (Frame.cpp:463-479): (cell_start[cols*rows+1], cell_items[n_in_grid])."""
    soa = np.ascontiguousarray(soa, np.int32)
    n = soa.size // 6
    start = np.zeros(cols * rows + 1, np.int32)
    items = np.zeros(max(n, 1), np.int32)
    k = lib(native).orc_assign_features_to_grid(n, soa.ctypes.data, min_x, min_y, inv_w, inv_h, cols, rows, start.ctypes.data, items.ctypes.data)
    return start, items[:k]
