"""Python host mirror of the reference's front-end interface, bound to libjsorb.so through its C ABI (include/jsorb.h).

Names and argument meaning follow the reference so that parity tests read like its call sites:
  ORBExtractor(...)            <- Jetson_SLAM::ORBExtractor::ORBExtractor       include/ORBextractor.h:25-35
  ORBExtractor.extract(image)  <- ORBExtractor::extract -> ORB_GPU::extract     include/ORBextractor.h:40-42, src/cuda/orb_gpu.cpp:489
  get_levels/get_scale_factor/get_scale_factors/get_inverse_scale_factors/
  get_scale_sigma_squares/get_inverse_scale_sigma_squares                      include/ORBextractor.h:44-72, src/ORBextractor.cpp:43-71
  compute_stereo_matches(l, r) <- Frame::ComputeStereoMatches                   src/Frame.cpp:780-803
There is no CPU fallback here: if libjsorb.so or a gfx950 device is missing the calls raise.
"""
import ctypes as C
import os

# the library's default for the HIP runtime (csrc/jsorb_api.hip, jsorb_runtime_defaults); effective when nothing has touched the GPU yet
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjsorb.so")

MAX_LEVELS = 16
TH_HIGH, TH_LOW = 100, 50   # ORBmatcher::TH_HIGH / TH_LOW, src/ORBmatcher.cpp:24-25

KERNELS = ["k_pyramid", "k_detect", "k_compact", "k_blur", "k_describe", "k_stereo", "k_median", "k_nms_ms"]

EXPORTS = [
    "jsorb_create", "jsorb_destroy", "jsorb_last_error", "jsorb_version", "jsorb_plan_launch", "jsorb_extract", "jsorb_extract_into", "jsorb_extract_device",
    "jsorb_extract_batch_device_async", "jsorb_extract_batch_host_async", "jsorb_sync", "jsorb_n_images", "jsorb_n_keypoints",
    "jsorb_level_n_keypoints", "jsorb_keypoints_device", "jsorb_descriptors_device", "jsorb_copy_keypoints",
    "jsorb_copy_descriptors", "jsorb_n_levels", "jsorb_level_dims", "jsorb_level_tiles", "jsorb_total_tiles", "jsorb_scale",
    "jsorb_inv_scale", "jsorb_level_image_device", "jsorb_copy_level_image", "jsorb_copy_tile_candidates", "jsorb_copy_angles",
    "jsorb_stereo_match", "jsorb_stereo_match_batch_async", "jsorb_stereo_uright_device", "jsorb_stereo_depth_device",
    "jsorb_copy_stereo", "jsorb_copy_stereo_l1", "jsorb_set_stereo_diagnostics", "jsorb_copy_stereo_diagnostics", "jsorb_set_speculative_stereo", "jsorb_speculative_stereo_stats", "jsorb_gather_counts_async", "jsorb_set_stream", "jsorb_get_stream", "jsorb_stream_wait_done", "jsorb_enable_kernel_timing", "jsorb_kernel_time",
    "jsorb_reset_kernel_timing", "jsorb_kernel_name", "jsorb_project_points", "jsorb_hamming_pairs", "jsorb_is_in_frustum",
    "jsorb_unpack_frame", "jsorb_assign_features_to_grid", "jsorb_copy_level_mask",
    "jsorb_mem_set_device", "jsorb_mem_alloc_host", "jsorb_mem_alloc_device", "jsorb_mem_alloc_device_pitched", "jsorb_mem_free_host",
    "jsorb_mem_free_device", "jsorb_mem_stream_create", "jsorb_mem_stream_destroy", "jsorb_mem_stream_sync", "jsorb_mem_device_sync", "jsorb_mem_buffer_sync", "jsorb_mem_h2d", "jsorb_mem_d2h",
    "jsorb_mem_d2d", "jsorb_mem_h2d_async", "jsorb_mem_d2h_async", "jsorb_mem_d2d_async", "jsorb_mem_set_zero", "jsorb_mem_set_zero_async",
    "jsorb_mem_last_error", "jsorb_read_mask_image", "jsorb_mask_image_last_error", "jsorb_create_masked",
]


# memory layout of cv::KeyPoint (jsorb_keypoint in include/jsorb.h)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class JsorbParams(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("n_levels", C.c_int), ("scale_factor", C.c_float),
                ("fast_n_min", C.c_int), ("fast_n_max", C.c_int), ("th_fast_min", C.c_int), ("th_fast_max", C.c_int),
                ("tile_h", C.c_int), ("tile_w", C.c_int), ("fixed_multi_scale_tile_size", C.c_int),
                ("apply_nms_ms", C.c_int), ("nms_ms_mode_gpu", C.c_int), ("device_id", C.c_int), ("max_batch", C.c_int)]


class JsorbStereoStats(C.Structure):
    _fields_ = [("n_left", C.c_int), ("n_right", C.c_int), ("n_candidate_pairs", C.c_int), ("n_corr_match", C.c_int),
                ("n_depth", C.c_int), ("n_final", C.c_int)]


class JsorbError(RuntimeError):
    pass


_lib = None
_libs_by_path = {}


def load_library(path=None):
    """dlopen libjsorb.so (the in-tree build).  Raises if it has not been built: there is no fallback path.
    With an explicit path: another build of the same ABI (jetson_slam_amd/build.py VARIANTS), loaded next to the default one and returned
    without replacing it - tests that need it swap the module's `_lib` for their duration."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    explicit = path is not None
    path = path or os.environ.get("JSORB_LIBRARY") or _LIB_PATH      # JSORB_LIBRARY: an alternative build of the same ABI
    if path in _libs_by_path:
        if not explicit:
            _lib = _libs_by_path[path]
        return _libs_by_path[path]
    if not os.path.exists(path):
        raise JsorbError("%s not found - run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)" % path)
    lib = C.CDLL(path)
    P, I, F = C.c_void_p, C.c_int, C.c_float
    sig = {
        "jsorb_create": (I, [C.POINTER(JsorbParams), P, C.POINTER(P)]),
        "jsorb_create_masked": (I, [C.POINTER(JsorbParams), P, I, I, C.POINTER(P)]),
        "jsorb_destroy": (None, [P]),
        "jsorb_last_error": (C.c_char_p, [P]),
        "jsorb_version": (C.c_char_p, []),
        "jsorb_plan_launch": (I, [C.POINTER(JsorbParams), P, I]),
        "jsorb_extract": (I, [P, P, I, C.POINTER(I)]),
        "jsorb_extract_into": (I, [P, P, I, C.POINTER(I), P, P]),
        "jsorb_extract_device": (I, [P, P, I, C.POINTER(I)]),
        "jsorb_extract_batch_device_async": (I, [P, P, C.c_size_t, I, I]),
        "jsorb_extract_batch_host_async": (I, [P, P, C.c_size_t, I, I]),
        "jsorb_sync": (I, [P]),
        "jsorb_n_images": (I, [P]),
        "jsorb_n_keypoints": (I, [P, I]),
        "jsorb_level_n_keypoints": (I, [P, I, I]),
        "jsorb_keypoints_device": (P, [P, I]),
        "jsorb_descriptors_device": (P, [P, I]),
        "jsorb_copy_keypoints": (I, [P, I, P]),
        "jsorb_copy_descriptors": (I, [P, I, P]),
        "jsorb_n_levels": (I, [P]),
        "jsorb_level_dims": (I, [P, I, C.POINTER(I), C.POINTER(I), C.POINTER(I)]),
        "jsorb_level_tiles": (I, [P, I, C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I), C.POINTER(I)]),
        "jsorb_total_tiles": (I, [P]),
        "jsorb_scale": (F, [P, I]),
        "jsorb_inv_scale": (F, [P, I]),
        "jsorb_level_image_device": (P, [P, I, I, I]),
        "jsorb_copy_level_image": (I, [P, I, I, I, P]),
        "jsorb_copy_level_mask": (I, [P, I, P]),
        "jsorb_read_mask_image": (I, [C.c_char_p, C.POINTER(I), C.POINTER(I), P, C.c_size_t]),
        "jsorb_mask_image_last_error": (C.c_char_p, []),
        "jsorb_copy_tile_candidates": (I, [P, I, P, P, P]),
        "jsorb_copy_angles": (I, [P, I, P]),
        "jsorb_stereo_match": (I, [P, P, F, F, I, I, P, P, C.POINTER(JsorbStereoStats)]),
        "jsorb_stereo_match_batch_async": (I, [P, P, F, F, I, I]),
        "jsorb_stereo_uright_device": (P, [P, I]),
        "jsorb_stereo_depth_device": (P, [P, I]),
        "jsorb_copy_stereo": (I, [P, I, P, P, C.POINTER(JsorbStereoStats)]),
        "jsorb_gather_counts_async": (I, [P, P, P]),
        "jsorb_copy_stereo_l1": (I, [P, I, P]),
        "jsorb_set_stereo_diagnostics": (I, [P, I]),
        "jsorb_copy_stereo_diagnostics": (I, [P, I, P]),
        "jsorb_set_speculative_stereo": (I, [P, I]),
        "jsorb_speculative_stereo_stats": (I, [P, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "jsorb_set_stream": (I, [P, P]),
        "jsorb_get_stream": (P, [P]),
        "jsorb_stream_wait_done": (I, [P, P]),
        "jsorb_enable_kernel_timing": (I, [P, I]),
        "jsorb_kernel_time": (I, [P, I, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
        "jsorb_reset_kernel_timing": (I, [P]),
        "jsorb_kernel_name": (C.c_char_p, [I]),
        "jsorb_project_points": (I, [P, I] + [P] * 5 + [F] * 8 + [P] * 4),
        "jsorb_hamming_pairs": (I, [P, I] + [P] * 5),
        "jsorb_is_in_frustum": (I, [P, I] + [P] * 12 + [F] * 4 + [I] * 5 + [F] * 2 + [P] * 6),
        "jsorb_unpack_frame": (I, [P, I, P, P]),
        "jsorb_assign_features_to_grid": (I, [P, I] + [F] * 4 + [I] * 2 + [P] * 2),
    }
    for name, (rt, at) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = rt, at
    _libs_by_path[path] = lib
    if not explicit:
        _lib = lib
    return lib


def read_mask_image(path):
    """cv::imread(path) + cvtColor(BGR2GRAY) of the reference's mask loading (orb_gpu.cpp:64-75) without OpenCV: PNG / binary PGM / PPM -> (H, W)
    uint8.  Returns None when the file cannot be opened (the reference then runs without a mask); raises on an unsupported format."""
    lib = load_library()
    w, h = C.c_int(), C.c_int()
    rc = lib.jsorb_read_mask_image(os.fsencode(path), C.byref(w), C.byref(h), None, 0)
    if rc == -4:
        return None
    if rc != 0:
        raise JsorbError("jsorb_read_mask_image rc=%d: %s" % (rc, lib.jsorb_mask_image_last_error().decode()))
    out = np.zeros((h.value, w.value), np.uint8)
    rc = lib.jsorb_read_mask_image(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data, out.size)
    if rc != 0:
        raise JsorbError("jsorb_read_mask_image rc=%d: %s" % (rc, lib.jsorb_mask_image_last_error().decode()))
    return out


def plan_launch(im_height, im_width, scale_factor, n_levels, tile_h=30, tile_w=30, fixed_multi_scale_tile_size=False, max_batch=1,
                FAST_N_MIN=9, FAST_N_MAX=14, th_FAST_MAX=20):
    """jsorb_plan_launch: the launch plan of a handle with these parameters, computed on the host (no GPU needed).  Returns a dict with the
    launch-wide numbers and a list of per-level dicts."""
    lib = load_library()
    prm = JsorbParams(im_height, im_width, n_levels, scale_factor, FAST_N_MIN, FAST_N_MAX, 7, th_FAST_MAX, tile_h, tile_w,
                      int(fixed_multi_scale_tile_size), 0, 0, 0, max_batch)
    out = np.zeros(8 + 8 * 16, np.int32)
    rc = lib.jsorb_plan_launch(C.byref(prm), out.ctypes.data, out.size)
    if rc != 0:
        raise JsorbError("jsorb_plan_launch rc=%d" % rc)
    keys = ("levels", "compact", "detect_lds", "spill_chunks", "pyramid_lds", "detect_blocks", "spill_chunk_entries", "reserved")
    res = dict(zip(keys, (int(v) for v in out[:8])))
    lk = ("det_R", "k_tiles", "pool", "score_stride", "list_cap", "pyr_ns16", "pyr_ns_dispatched", "tile_rows")
    res["per_level"] = [dict(zip(lk, (int(v) for v in out[8 + 8 * i:16 + 8 * i]))) for i in range(res["levels"])]
    return res


class ORBExtractor:
    """Mirror of Jetson_SLAM::ORBExtractor (include/ORBextractor.h:21-93) over the C ABI.

    Argument order and meaning are the reference constructor's; `use_gpu` is accepted and ignored exactly as the
    reference does (src/ORBextractor.cpp:75-87 always builds the GPU object).  `max_batch` and `device_id` are additions.
    """

    def __init__(self, im_height, im_width, scale_factor, n_levels, FAST_N_MIN, FAST_N_MAX, th_FAST_MIN, th_FAST_MAX,
                 str_mask=None, tile_h=30, tile_w=30, fixed_multi_scale_tile_size=False, apply_nms_ms=False,
                 nms_ms_mode_gpu=False, use_gpu=True, device_id=0, max_batch=1):
        self._lib = load_library()
        self._h = C.c_void_p()
        mask = None
        if isinstance(str_mask, str) and str_mask:
            # the reference's image path (None = unreadable = no mask, orb_gpu.cpp:69-73); whatever size the image has, every level is
            # resized from it directly with INTER_NN (orb_gpu.cpp:77-81): it goes through the ABI at its own size
            mask = read_mask_image(str_mask)
        elif str_mask is not None and not isinstance(str_mask, str):
            mask = np.ascontiguousarray(str_mask, np.uint8)   # a 2-D array (any size) instead of the reference's image path
            assert mask.ndim == 2
        self.params = JsorbParams(im_height, im_width, n_levels, scale_factor, FAST_N_MIN, FAST_N_MAX, th_FAST_MIN,
                                  th_FAST_MAX, tile_h, tile_w, int(fixed_multi_scale_tile_size), int(apply_nms_ms),
                                  int(nms_ms_mode_gpu), device_id, max_batch)
        if mask is None:
            rc = self._lib.jsorb_create(C.byref(self.params), None, C.byref(self._h))
        else:
            rc = self._lib.jsorb_create_masked(C.byref(self.params), mask.ctypes.data, int(mask.shape[1]), int(mask.shape[0]), C.byref(self._h))
        if rc != 0:
            msg = self._lib.jsorb_last_error(self._h).decode() if self._h else "jsorb_create failed"
            if self._h:
                self._lib.jsorb_destroy(self._h)
                self._h = C.c_void_p()
            raise JsorbError("jsorb_create rc=%d: %s" % (rc, msg))
        self.n_levels_ = n_levels
        self.scale_factor_ = np.float32(scale_factor)
        # scale tables for the SLAM side (src/ORBextractor.cpp:43-71), float32 arithmetic as in the reference
        s = np.ones(n_levels, np.float32)
        for i in range(1, n_levels):
            s[i] = np.float32(s[i - 1] * self.scale_factor_)
        self.scale_ = s
        self.level_sigma2_ = (s * s).astype(np.float32)
        self.level_sigma2_[0] = np.float32(1.0)
        self.inv_scale_ = (np.float32(1.0) / s).astype(np.float32)
        self.inv_level_sigma2_ = (np.float32(1.0) / self.level_sigma2_).astype(np.float32)
        self.T = self._lib.jsorb_total_tiles(self._h)
        self.max_batch = max_batch
        self._keep = None

    # ---- reference getters ----
    def get_levels(self):
        return self.n_levels_

    def get_scale_factor(self):
        return float(self.scale_factor_)

    def get_scale_factors(self):
        return self.scale_.copy()

    def get_inverse_scale_factors(self):
        return self.inv_scale_.copy()

    def get_scale_sigma_squares(self):
        return self.level_sigma2_.copy()

    def get_inverse_scale_sigma_squares(self):
        return self.inv_level_sigma2_.copy()

    # ---- helpers ----
    def _chk(self, rc):
        if rc != 0:
            raise JsorbError("libjsorb rc=%d: %s" % (rc, self._lib.jsorb_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.jsorb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # ---- extraction ----
    def extract(self, image):
        """ORBExtractor::extract: one host image -> (keypoints int32[6N] SoA, descriptors uint8[N,32]) pulled to the host
        (the reference leaves them in SyncedMem and the Frame ctor calls to_cpu(), Frame.cpp:119-122)."""
        image = np.ascontiguousarray(image, np.uint8)
        assert image.shape == (self.params.height, self.params.width)
        n = C.c_int()
        self._chk(self._lib.jsorb_extract(self._h, image.ctypes.data, image.strides[0], C.byref(n)))
        return self.keypoints(0), self.descriptors(0)

    def extract_batch_device_async(self, dev_ptr, image_stride, step, n_images, keep=None):
        """Batch mode on images already resident in HBM (raw device pointer). `keep` pins a Python owner of the memory."""
        self._keep = keep
        self._chk(self._lib.jsorb_extract_batch_device_async(self._h, dev_ptr, image_stride, step, n_images))

    def extract_batch_host_async(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        assert images.ndim == 3 and images.shape[1:] == (self.params.height, self.params.width)
        self._keep = images
        self._chk(self._lib.jsorb_extract_batch_host_async(self._h, images.ctypes.data, images.strides[0], images.strides[1],
                                                           images.shape[0]))

    def sync(self):
        self._chk(self._lib.jsorb_sync(self._h))

    def n_keypoints(self, image=0):
        return self._lib.jsorb_n_keypoints(self._h, image)

    def level_n_keypoints(self, image=0):
        return [self._lib.jsorb_level_n_keypoints(self._h, image, l) for l in range(self.n_levels_)]

    def keypoints(self, image=0):
        n = self.n_keypoints(image)
        if n < 0:
            raise JsorbError("no extract result for image %d" % image)
        out = np.zeros(6 * n, np.int32)
        if n:
            self._chk(self._lib.jsorb_copy_keypoints(self._h, image, out.ctypes.data))
        return out

    def descriptors(self, image=0):
        n = self.n_keypoints(image)
        out = np.zeros((n, 32), np.uint8)
        if n:
            self._chk(self._lib.jsorb_copy_descriptors(self._h, image, out.ctypes.data))
        return out

    def unpack_frame(self, image=0):
        """Frame.cpp:119-196 on the device: (keypoints as a structured array with the layout of cv::KeyPoint, descriptors N x 32)."""
        n = self.n_keypoints(image)
        if n < 0:
            raise JsorbError("no extract result for image %d" % image)
        kps = np.zeros(n, KEYPOINT_DTYPE)
        desc = np.zeros((n, 32), np.uint8)
        if n:
            self._chk(self._lib.jsorb_unpack_frame(self._h, image, kps.ctypes.data, desc.ctypes.data))
        return kps, desc

    def assign_features_to_grid(self, min_x, min_y, grid_element_width_inv, grid_element_height_inv, cols=64, rows=48, image=0):
        """Frame::AssignFeaturesToGrid (Frame.cpp:463-479) as CSR: (cell_start[cols*rows+1], cell_items); cell (i, j) = i*rows + j."""
        n = self.n_keypoints(image)
        if n < 0:
            raise JsorbError("no extract result for image %d" % image)
        start = np.zeros(cols * rows + 1, np.int32)
        items = np.zeros(max(n, 1), np.int32)
        self._chk(self._lib.jsorb_assign_features_to_grid(self._h, image, min_x, min_y, grid_element_width_inv, grid_element_height_inv,
                                                          cols, rows, start.ctypes.data, items.ctypes.data))
        return start, items[:start[-1]]

    def angles(self, image=0):
        n = self.n_keypoints(image)
        out = np.zeros(n, np.float32)
        if n:
            self._chk(self._lib.jsorb_copy_angles(self._h, image, out.ctypes.data))
        return out

    def level_dims(self):
        res = []
        for l in range(self.n_levels_):
            h, w, p = C.c_int(), C.c_int(), C.c_int()
            self._chk(self._lib.jsorb_level_dims(self._h, l, C.byref(h), C.byref(w), C.byref(p)))
            res.append((h.value, w.value))
        return res

    def level_tiles(self):
        res = []
        for l in range(self.n_levels_):
            v = [C.c_int() for _ in range(5)]
            self._chk(self._lib.jsorb_level_tiles(self._h, l, *[C.byref(t) for t in v]))
            res.append(tuple(t.value for t in v))   # (tile_h, tile_w, n_tile_h, n_tile_w, level_offset)
        return res

    def level_image(self, level, image=0, blurred=False):
        h, w = self.level_dims()[level]
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.jsorb_copy_level_image(self._h, image, level, int(blurred), out.ctypes.data))
        return out

    def level_mask(self, level):
        """ORB_GPU::masks_[level] (orb_gpu.cpp:64-91): 0 / 255 plane of the level"""
        h, w = self.level_dims()[level]
        out = np.zeros((h, w), np.uint8)
        self._chk(self._lib.jsorb_copy_level_mask(self._h, level, out.ctypes.data))
        return out

    def tile_candidates(self, image=0):
        x, y, s = (np.zeros(self.T, np.int32) for _ in range(3))
        self._chk(self._lib.jsorb_copy_tile_candidates(self._h, image, x.ctypes.data, y.ctypes.data, s.ctypes.data))
        return x, y, s

    # ---- profiling plumbing ----
    def set_stream(self, stream_ptr):
        self._chk(self._lib.jsorb_set_stream(self._h, stream_ptr))

    def stream_wait_done(self, other_stream_ptr):
        """make another HIP stream (raw pointer, 0 = null stream) wait for everything enqueued on this handle"""
        self._chk(self._lib.jsorb_stream_wait_done(self._h, other_stream_ptr))

    def enable_kernel_timing(self, on=True):
        self._chk(self._lib.jsorb_enable_kernel_timing(self._h, int(on)))

    def reset_kernel_timing(self):
        self._chk(self._lib.jsorb_reset_kernel_timing(self._h))

    def kernel_times(self):
        """{kernel: (total_ms, launches)} measured with hipEvents on the stream the kernels run on."""
        res = {}
        for i, name in enumerate(KERNELS):
            ms, n = C.c_double(), C.c_long()
            self._chk(self._lib.jsorb_kernel_time(self._h, i, C.byref(ms), C.byref(n)))
            res[name] = (ms.value, n.value)
        return res


def compute_stereo_matches(left, right, mb, mbf, th_high=TH_HIGH, th_low=TH_LOW):
    """Frame::ComputeStereoMatches (src/Frame.cpp:780-803) on the last extract of two ORBExtractor objects.
    Returns (mvuRight, mvDepth, stats) as float32 arrays of length N_left (-1 = no match)."""
    n = left.n_keypoints(0)
    u = np.full(max(n, 1), -1, np.float32)
    d = np.full(max(n, 1), -1, np.float32)
    st = JsorbStereoStats()
    rc = left._lib.jsorb_stereo_match(left.handle, right.handle, mb, mbf, th_high, th_low, u.ctypes.data, d.ctypes.data, C.byref(st))
    left._chk(rc)
    return u[:n], d[:n], {k: getattr(st, k) for k, _ in JsorbStereoStats._fields_}


def set_speculative_stereo(left, on):
    """jsorb_set_speculative_stereo: the match enqueued behind the next pair of single-image extracts (include/jsorb.h)"""
    left._chk(left._lib.jsorb_set_speculative_stereo(left.handle, int(bool(on))))


def speculative_stereo_stats(left):
    """(adopted, dropped) speculative matches of the pair this left handle belongs to"""
    a, d = C.c_long(0), C.c_long(0)
    left._chk(left._lib.jsorb_speculative_stereo_stats(left.handle, C.byref(a), C.byref(d)))
    return a.value, d.value


def stereo_match_batch_async(left, right, mb, mbf, th_high=TH_HIGH, th_low=TH_LOW):
    left._chk(left._lib.jsorb_stereo_match_batch_async(left.handle, right.handle, mb, mbf, th_high, th_low))


def stereo_result(left, image=0):
    n = left.n_keypoints(image)
    u = np.full(max(n, 1), -1, np.float32)
    d = np.full(max(n, 1), -1, np.float32)
    st = JsorbStereoStats()
    left._chk(left._lib.jsorb_copy_stereo(left.handle, image, u.ctypes.data, d.ctypes.data, C.byref(st)))
    return u[:n], d[:n], {k: getattr(st, k) for k, _ in JsorbStereoStats._fields_}


def stereo_l1(left, image=0):
    """L1 distances the median cut sorted (-1 = no accepted refinement), int32[N_left]"""
    n = left.n_keypoints(image)
    out = np.full(max(n, 1), -1, np.int32)
    left._chk(left._lib.jsorb_copy_stereo_l1(left.handle, image, out.ctypes.data))
    return out[:n]


def set_stereo_diagnostics(left, on=True):
    """keep the matcher's intermediate results of the following matches (stereo_diagnostics)"""
    left._chk(left._lib.jsorb_set_stereo_diagnostics(left.handle, 1 if on else 0))


def stereo_diagnostics(left, image=0):
    """(best_right[N], best_dist[N], l1_sums[N, 11]) of the last match: K12's arg-min per left keypoint (-1 / th_high: none) and the 11 L1 window
    sums of K13 (-1 where no window search ran) - what the reference keeps on the host between its two kernels"""
    n = left.n_keypoints(image)
    out = np.full((max(n, 1), 13), -1, np.int32)
    left._chk(left._lib.jsorb_copy_stereo_diagnostics(left.handle, image, out.ctypes.data))
    out = out[:n]
    return out[:, 0].copy(), out[:, 1].copy(), out[:, 2:].copy()


def gather_counts_async(left, right, dev_dst_ptr):
    """(N_left, N_right, N_matched) per pair of the last batch -> int32[3*n] DEVICE buffer (payload of the RCCL all_gather)."""
    left._chk(left._lib.jsorb_gather_counts_async(left.handle, right.handle, dev_dst_ptr))
