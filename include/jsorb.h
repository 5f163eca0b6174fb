/*
 * jsorb.h - C ABI of libjsorb: the MI355X-native (HIP, gfx950) ORB front-end + stereo matcher that
 * replaces Jetson-SLAM's CUDA hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * Reference interfaces replaced (paths under the reference tree, ashishkumar822/Jetson-SLAM @ 2024-12-18):
 *   jsorb_create / jsorb_destroy   <- orb_cuda::ORB_GPU::ORB_GPU / ~ORB_GPU   include/cuda/orb_gpu.hpp:26-36, src/cuda/orb_gpu.cpp:22-451
 *                                     (built by Jetson_SLAM::ORBExtractor::ORBExtractor, include/ORBextractor.h:25-35, src/ORBextractor.cpp:75-87)
 *   jsorb_extract*                 <- ORBExtractor::extract -> ORB_GPU::extract   include/ORBextractor.h:40-42, src/cuda/orb_gpu.cpp:489-841
 *   jsorb_keypoints_* / jsorb_descriptors_* <- SyncedMem<int>/<unsigned char>::gpu_data()/to_cpu()  include/cuda/synced_mem_holder.hpp:10-65 (Frame.cpp:119-122)
 *   jsorb_level_* / jsorb_scale_*  <- public members height_, width_, image_, scale_, inv_scale_  include/cuda/orb_gpu.hpp:240-250 (used by Frame.cpp:784-801)
 *   jsorb_stereo_match*            <- Frame::ComputeStereoMatches -> ORB_GPU::ORB_compute_stereo_match  src/Frame.cpp:780-803, include/cuda/orb_gpu.hpp:218-229,
 *                                     src/cuda/orb_stereo_match.cu:105-580
 *
 * Output layout is the reference's: keypoints = 6N int32 in six consecutive blocks x[N] y[N] score[N] angle[N] (degrees, f32 bit
 * pattern) octave[N] size[N]; descriptors = 32N bytes; keypoint order = level-major, tile-raster within a level.
 * All functions return JSORB_OK (0) or a negative error; nothing throws across this boundary.
 * Threading: distinct handles may be used concurrently from different host threads (the reference runs the left and right
 * extractors in two std::threads, Frame.cpp:107-110); one handle is single-threaded.  There is no CPU fallback: if no gfx950
 * device / code object is available the calls fail with JSORB_ERR_HIP.
 */
#ifndef JSORB_H
#define JSORB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSORB_OK 0
#define JSORB_ERR_INVALID (-1)      /* bad argument / unsupported parameter combination */
#define JSORB_ERR_HIP (-2)          /* a HIP runtime call failed (see jsorb_last_error) */
#define JSORB_ERR_UNSUPPORTED (-3)  /* parameter combination outside what this build supports */
#define JSORB_ERR_STATE (-4)        /* call order violation (e.g. stereo before extract) */

#define JSORB_MAX_LEVELS 16

typedef struct jsorb_extractor jsorb_extractor;

/* Mirrors the ORB_GPU / ORBExtractor constructor arguments (include/cuda/orb_gpu.hpp:26-36). */
typedef struct jsorb_params {
    int height, width;               /* level-0 image size */
    int n_levels;                    /* ORBextractor.nLevels */
    float scale_factor;              /* ORBextractor.scaleFactor */
    int fast_n_min, fast_n_max;      /* ORBextractor.FAST_N_MIN / FAST_N_MAX : bounded arc length */
    int th_fast_min, th_fast_max;    /* th_FAST_MIN is accepted and ignored exactly as the reference does (orb_gpu.cpp:42-47) */
    int tile_h, tile_w;              /* ORBextractor.tile_h / tile_w (level 0) */
    int fixed_multi_scale_tile_size;
    int apply_nms_ms, nms_ms_mode_gpu;
    int device_id;                   /* reference hard-wires 0 (ORBextractor.cpp:87) */
    int max_batch;                   /* images one extract call may process (>=1); 1 = reference behaviour - such a handle lays its launches out for the
                                        latency of ONE image (many short workgroups), a handle with max_batch > 1 for throughput */
} jsorb_params;

typedef struct jsorb_stereo_stats {
    int n_left, n_right;
    int n_candidate_pairs;           /* (iL,iR) pairs whose Hamming distance was evaluated */
    int n_corr_match;                /* matches refined by the 11x11 L1 window search */
    int n_depth;                     /* matches with a depth before the median cut */
    int n_final;                     /* after the 2.1 x median cut */
} jsorb_stereo_stats;

/* kernel ids for jsorb_kernel_time */
enum { JSORB_K_PYRAMID = 0, JSORB_K_DETECT, JSORB_K_COMPACT, JSORB_K_BLUR, JSORB_K_DESCRIBE, JSORB_K_STEREO, JSORB_K_MEDIAN, JSORB_K_NMS_MS, JSORB_K_COUNT };

/* ---- lifetime ---- */
/* mask: NULL (no mask => all 255) or a height*width u8 level-0 mask in host memory. */
int jsorb_create(const jsorb_params *params, const uint8_t *mask, jsorb_extractor **out);
/* The mask at ITS OWN size (mask_width x mask_height, any size): every level, level 0 included, is resized from it directly with
 * cv::resize(INTER_NN)'s index rule - what orb_gpu.cpp:77-81 does with whatever image the yaml names.  jsorb_create is this call with
 * the level-0 size. */
int jsorb_create_masked(const jsorb_params *params, const uint8_t *mask, int mask_width, int mask_height, jsorb_extractor **out);
void jsorb_destroy(jsorb_extractor *e);
const char *jsorb_last_error(const jsorb_extractor *e);
const char *jsorb_version(void);
/* Host-only, touches no device: the launch plan a handle created with these parameters gets (no counterpart in the reference, whose launch shapes
 * are literals in its .cu files, e.g. src/cuda/orb_FAST_apply_NMS_G.cu:1405-1434).  out[0..7] = levels, k_detect form (1: compact, 0: full plane),
 * k_detect LDS bytes, spill chunks in the handle's arena, k_pyramid LDS bytes, k_detect workgroups per image, entries per spill chunk, 0 (reserved);
 * then 8 ints per level: tile rows per k_detect workgroup, tiles per workgroup, LDS pool entries, score-plane stride, survivor-list capacity,
 * 16-byte loads per lane and row of k_pyramid, the load count its kernel instantiates for that, tile rows of the level.  capacity >= 8 + 8 * levels. */
int jsorb_plan_launch(const jsorb_params *params, int32_t *out, int capacity);

/* The mask image the reference loads with cv::imread(str_mask) + cvtColor(BGR2GRAY) (orb_gpu.cpp:64-75; yaml keys mask.left / mask.right):
 * decodes a PNG (non-interlaced; gray, gray+alpha, RGB, RGBA, palette; 1-16 bit) or a binary PGM / PPM file to one gray byte per pixel, for
 * callers that do not link OpenCV.  Call with gray_out = NULL to obtain the size first.  JSORB_ERR_STATE: the file cannot be opened (the
 * reference then runs without a mask); JSORB_ERR_UNSUPPORTED: not one of these formats (text in jsorb_mask_image_last_error). */
int jsorb_read_mask_image(const char *path, int *width, int *height, uint8_t *gray_out, size_t capacity);
const char *jsorb_mask_image_last_error(void);

/* ---- extraction ---- */
/* Reference-shaped call: one host image (step = bytes per row), results stay on the device; *n_keypoints = N.  Synchronous. */
int jsorb_extract(jsorb_extractor *e, const uint8_t *host_image, int step, int *n_keypoints);
/* Same, and the results are ALSO delivered into caller-owned device buffers (6*T int32 and 32*T bytes, T = jsorb_total_tiles, the
 * keypoint cap) in the same stream round trip: what the reference's extract() does with the caller's SyncedMem (orb_gpu.cpp:779-831
 * writes into out_keypoints.gpu_data() / out_keypoints_desc.gpu_data()).  Either destination may be NULL. */
int jsorb_extract_into(jsorb_extractor *e, const uint8_t *host_image, int step, int *n_keypoints, int32_t *dev_keypoints_dst, uint8_t *dev_descriptors_dst);
/* Same with the image already in device memory. */
int jsorb_extract_device(jsorb_extractor *e, const uint8_t *dev_image, int step, int *n_keypoints);
/* Batch mode: n_images (<= max_batch) images at dev_images + i*image_stride, rows `step` bytes apart.  Enqueues only; the
 * input buffer must stay valid until jsorb_sync (level 0 is read in place).  Results per image via the accessors below. */
int jsorb_extract_batch_device_async(jsorb_extractor *e, const uint8_t *dev_images, size_t image_stride, int step, int n_images);
/* Batch mode from host memory (pinned for asynchronous uploads; pageable works, the copy then blocks the caller).  Dense input
 * (image_stride == H*W, step == W, W a multiple of 16): one hipMemcpyAsync per batch on the device's upload stream into one of two
 * landing buffers that the kernels read in place, so that the upload of batch k+1 runs under the kernels of batch k; the host buffer may
 * be reused once the upload has run (jsorb_sync, or any later call that returned after it).  One image (n_images == 1) from any host
 * memory: copied into a pinned buffer of the handle by the calling thread and pulled over PCIe by the first kernel.  Strided input:
 * hipMemcpy2DAsync per image into the level-0 slab. */
int jsorb_extract_batch_host_async(jsorb_extractor *e, const uint8_t *host_images, size_t image_stride, int step, int n_images);
/* Wait for everything enqueued on this handle and refresh the host-side counts. */
int jsorb_sync(jsorb_extractor *e);

/* ---- results (valid after a synchronous call or jsorb_sync, until the next extract on the handle) ---- */
int jsorb_n_images(const jsorb_extractor *e);
int jsorb_n_keypoints(const jsorb_extractor *e, int image);
int jsorb_level_n_keypoints(const jsorb_extractor *e, int image, int level);
const int32_t *jsorb_keypoints_device(const jsorb_extractor *e, int image);   /* 6N int32, layout above */
const uint8_t *jsorb_descriptors_device(const jsorb_extractor *e, int image); /* 32N bytes */
int jsorb_copy_keypoints(const jsorb_extractor *e, int image, int32_t *host_dst /* 6N */);
int jsorb_copy_descriptors(const jsorb_extractor *e, int image, uint8_t *host_dst /* 32N */);

/* ---- geometry / tables (mirrors ORB_GPU::height_, width_, scale_, inv_scale_, tile grid) ---- */
int jsorb_n_levels(const jsorb_extractor *e);
int jsorb_level_dims(const jsorb_extractor *e, int level, int *height, int *width, int *pitch);
int jsorb_level_tiles(const jsorb_extractor *e, int level, int *tile_h, int *tile_w, int *n_tile_h, int *n_tile_w, int *level_offset);
int jsorb_total_tiles(const jsorb_extractor *e);
float jsorb_scale(const jsorb_extractor *e, int level);
float jsorb_inv_scale(const jsorb_extractor *e, int level);
/* Device pointer to the un-blurred (blurred=0) or 7x7-blurred (blurred=1) pyramid level of an image (ORB_GPU::image_/image_gaussian_). */
const uint8_t *jsorb_level_image_device(const jsorb_extractor *e, int image, int level, int blurred);
int jsorb_copy_level_image(const jsorb_extractor *e, int image, int level, int blurred, uint8_t *host_dst /* H*W, pitch W */);
/* The per-level feature mask (ORB_GPU::masks_[level], orb_gpu.cpp:64-91: cv::resize(INTER_NN) of the level-0 mask, then
 * threshold(10)): H*W bytes 0 / 255, pitch W; all 255 when the handle has no mask. */
int jsorb_copy_level_mask(const jsorb_extractor *e, int level, uint8_t *host_dst);
/* Per-tile candidates before compaction (x,y,score), T entries each: debugging / stage-level parity. */
int jsorb_copy_tile_candidates(const jsorb_extractor *e, int image, int32_t *x, int32_t *y, int32_t *score);
/* Per-keypoint orientation in radians in output order (N floats). */
int jsorb_copy_angles(const jsorb_extractor *e, int image, float *host_dst);

/* ---- stereo ---- */
/* Reference-shaped call on image 0 of each handle: fills u_right[N_left], depth[N_left] (host, -1 = no match).  Synchronous.
 * mb is passed explicitly (the reference reads Frame::mb before it is assigned - SURVEY Appendix C-5); th_high/th_low are
 * ORBmatcher::TH_HIGH/TH_LOW = 100/50 (ORBmatcher.cpp:24-25). */
int jsorb_stereo_match(jsorb_extractor *left, jsorb_extractor *right, float mb, float mbf, int th_high, int th_low,
                       float *u_right, float *depth, jsorb_stereo_stats *stats);
/* Speculative match: OFF unless asked for - jsorb_set_speculative_stereo(left, 1), which include/jsorb_compat.hpp calls when it sees the
 * Frame stereo call shape (ORB_compute_stereo_match / ComputeStereoMatches), or JSORB_SPECULATE=1; JSORB_SPECULATE=0 forbids it whatever
 * the caller asks for.  A mono / RGB-D flow (one extractor, no match) never arms it.  After one jsorb_stereo_match on a (left, right)
 * pair of single-image handles, the library enqueues the same match, with the same mb / mbf / thresholds, right behind the NEXT pair of
 * single-image extracts on the GPU (the extract call that arrives second does it), so that the following jsorb_stereo_match on that
 * pair finds its result finished instead of paying a host round trip between extract and match.  The result is adopted only when the
 * call names the same pair and parameters and neither handle has extracted again; otherwise it is dropped and the call runs as if
 * the feature did not exist.  Same kernels, same inputs, same outputs.  The two extract calls may come from two threads (as in
 * Frame.cpp:107-110); calls on ONE handle must not overlap, as before. */
int jsorb_set_speculative_stereo(jsorb_extractor *left, int on);
int jsorb_speculative_stereo_stats(const jsorb_extractor *left, long *n_adopted, long *n_dropped);
/* Batch mode: image i of `left` against image i of `right`; enqueues on left's stream after right's work. */
int jsorb_stereo_match_batch_async(jsorb_extractor *left, jsorb_extractor *right, float mb, float mbf, int th_high, int th_low);
const float *jsorb_stereo_uright_device(const jsorb_extractor *left, int image);
const float *jsorb_stereo_depth_device(const jsorb_extractor *left, int image);
int jsorb_copy_stereo(const jsorb_extractor *left, int image, float *u_right, float *depth, jsorb_stereo_stats *stats);
/* Inspection (tests): the L1 window distance of the accepted sub-pixel refinement per left keypoint, -1 = none - the values the median
 * cut of orb_stereo_match.cu:560-580 sorts (vDistIdx[].first), N_left int32. */
int jsorb_copy_stereo_l1(const jsorb_extractor *left, int image, int32_t *host_dst);
/* Inspection (tests): keep the intermediate results of the matcher the reference holds on the host between its two kernels -
 * per left keypoint 13 int32 = { best right index of K12's arg-min (-1: no candidate closer than th_high), its Hamming distance (th_high then),
 * the 11 L1 window sums of K13 + cublasSgemv (-1 where no window search ran) } - orb_stereo_match.cu:241-256, 294-470.  Off by default
 * (B x T x 52 bytes of device memory); while on, a speculative single-frame match is never adopted. */
int jsorb_set_stereo_diagnostics(jsorb_extractor *left, int on);
int jsorb_copy_stereo_diagnostics(const jsorb_extractor *left, int image, int32_t *host_dst /* 13 * N_left */);
/* Multi-GPU batch mode: write (N_left, N_right, N_matched) of every pair of the last batch, 3 int32 per pair, to a DEVICE
 * buffer (enqueued on left's stream) - the payload of the one collective of this path, an all-gather of per-pair counts. */
int jsorb_gather_counts_async(jsorb_extractor *left, jsorb_extractor *right, int32_t *dev_dst);

/* ---- Tracking-side GPU helpers: device pointers in and out, synchronous like the reference ---- */
/* orb_cuda::ORB_Search_by_projection_project_on_frame  include/cuda/orb_matcher.hpp:12-18, src/cuda/orb_matcher.cu:17-89 */
int jsorb_project_points(void *hip_stream, int n_points, const float *Px, const float *Py, const float *Pz, const float *Rcw, const float *tcw,
                         float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
                         float *u, float *v, float *invz, unsigned char *is_valid);
/* orb_cuda::ORB_compute_distances  include/cuda/orb_matcher.hpp:20-24, src/cuda/orb_matcher.cu:95-144 (descriptor bases 16-byte aligned) */
int jsorb_hamming_pairs(void *hip_stream, int n_pairs, const int *idx_left, const int *idx_right, const unsigned char *descriptor_left,
                        const unsigned char *descriptor_right, int *distance);
/* tracking_cuda::compute_isInFrustum_GPU  include/cuda/tracking_gpu.hpp, src/cuda/tracking_isinfrustum.cu:19-160
 * (u, v, invz, predictedlevel, viewCos are written only where is_infrustum becomes 1, as in the reference) */
int jsorb_is_in_frustum(void *hip_stream, int n_points, const float *Px, const float *Py, const float *Pz, const float *Pnx, const float *Pny,
                        const float *Pnz, const float *MaxDistance, const float *invariance_maxDistance, const float *invariance_minDistance,
                        const float *Rcw, const float *tcw, const float *Ow, float fx, float fy, float cx, float cy, int minX, int maxX, int minY,
                        int maxY, int nScaleLevels, float logScaleFactor, float viewCosAngle, float *invz, float *u, float *v, int *predictedlevel,
                        float *viewCos, unsigned char *is_infrustum);

/* ---- Frame-side unpacking: what Frame::Frame does on the host right after the two extract() calls ---- */
/* The memory layout of cv::KeyPoint (OpenCV 4: Point2f pt; float size, angle, response; int octave, class_id), 28 bytes. */
typedef struct jsorb_keypoint { float x, y, size, angle, response; int32_t octave, class_id; } jsorb_keypoint;
/* Frame.cpp:119-196: the keypoint SoA of one image of the last batch as cv::KeyPoint records + its descriptor rows (N x 32).
 * A device kernel interleaves the SoA; both arrays then come back with two asynchronous copies and ONE synchronisation (the
 * reference: four blocking SyncedMem::to_cpu() per stereo frame, then a host loop).  Either destination may be NULL. */
int jsorb_unpack_frame(jsorb_extractor *e, int image, jsorb_keypoint *keypoints, uint8_t *descriptors);
/* Frame::AssignFeaturesToGrid + Frame::PosInGrid (Frame.cpp:463-479, 696-706) on the device, as CSR over cols x rows cells:
 * cell (i, j) has index i*rows + j (mGrid[i][j]); cell_start has cols*rows + 1 entries; cell_items lists keypoint indices,
 * ascending inside a cell (the reference's push_back order).  Uses the extracted keypoint coordinates (mvKeysUn == mvKeys for
 * rectified stereo).  Host destinations; cols*rows <= 16384. */
int jsorb_assign_features_to_grid(jsorb_extractor *e, int image, float min_x, float min_y, float grid_element_width_inv,
                                  float grid_element_height_inv, int cols, int rows, int32_t *cell_start, int32_t *cell_items);

/* ---- memory: what orb_cuda::SyncedMem<T> needs (include/cuda/synced_mem_holder.hpp:10-65, src/cuda/synced_mem_holder.cpp:8-199) ----
 * The reference's untouched host code (ORBmatcher.cpp:1673-1877, Tracking.cpp:1427-1600, orb_stereo_match.cu statics) allocates
 * pinned-host + device buffer pairs and moves data with cudaMemcpy(Async) on a private stream; these calls are the HIP side of
 * that, so that include/jsorb_compat.hpp can offer the full SyncedMem surface without the consumer including a HIP header.
 * They act on the calling thread's current device (jsorb_mem_set_device), like the CUDA runtime calls they replace.
 * `stream` is a hipStream_t as void*; NULL = the null stream (blocking calls) / synchronous copy. */
int jsorb_mem_set_device(int device_id);
int jsorb_mem_alloc_host(size_t bytes, void **host_pinned);                  /* cudaMallocHost  (synced_mem_holder.cpp:63) */
int jsorb_mem_alloc_device(size_t bytes, void **device);                     /* cudaMalloc      (:67) */
int jsorb_mem_alloc_device_pitched(size_t width_bytes, size_t height, void **device, size_t *pitch);   /* cudaMallocPitch (:50) */
int jsorb_mem_free_host(void *host_pinned);                                  /* cudaFreeHost */
int jsorb_mem_free_device(void *device);                                     /* cudaFree */
int jsorb_mem_stream_create(void **stream);                                  /* cudaStreamCreate (:19) */
int jsorb_mem_stream_destroy(void *stream);
int jsorb_mem_stream_sync(void *stream);                                     /* cudaStreamSynchronize (:190) */
int jsorb_mem_device_sync(void);                                             /* the device-wide wait cudaFree / cudaFreeHost imply (:31-44): before a buffer is recycled */
int jsorb_mem_buffer_sync(const void *device);                               /* the same wait on the device that owns `device` (cudaFree waits for the buffer's device, not the caller's current one) */
int jsorb_mem_h2d(void *device_dst, const void *host_src, size_t bytes);     /* cudaMemcpy HostToDevice (:96) */
int jsorb_mem_d2h(void *host_dst, const void *device_src, size_t bytes);     /* cudaMemcpy DeviceToHost (:90) */
int jsorb_mem_d2d(void *device_dst, const void *device_src, size_t bytes);
int jsorb_mem_h2d_async(void *device_dst, const void *host_src, size_t bytes, void *stream);   /* cudaMemcpyAsync (:128) */
int jsorb_mem_d2h_async(void *host_dst, const void *device_src, size_t bytes, void *stream);   /* cudaMemcpyAsync (:122) */
int jsorb_mem_d2d_async(void *device_dst, const void *device_src, size_t bytes, void *stream);
int jsorb_mem_set_zero(void *device, size_t bytes);                          /* cudaMemset (:109) */
int jsorb_mem_set_zero_async(void *device, size_t bytes, void *stream);      /* cudaMemsetAsync (:115) */
const char *jsorb_mem_last_error(void);                                      /* text of the last failed jsorb_mem_* call of this thread */

/* ---- plumbing ---- */
/* Use an external HIP stream (hipStream_t as void*) instead of the handle's own, e.g. torch's current stream. NULL restores. */
int jsorb_set_stream(jsorb_extractor *e, void *hip_stream);
void *jsorb_get_stream(const jsorb_extractor *e);
/* Make another HIP stream (e.g. the one a collective will be issued on) wait for everything enqueued so far on this handle. */
int jsorb_stream_wait_done(jsorb_extractor *e, void *other_hip_stream);
/* Per-kernel hipEvent timing (off by default: it serialises launches). Accumulates until reset. */
int jsorb_enable_kernel_timing(jsorb_extractor *e, int on);
int jsorb_kernel_time(jsorb_extractor *e, int kernel_id, double *total_ms, long *launches);
int jsorb_reset_kernel_timing(jsorb_extractor *e);
const char *jsorb_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* JSORB_H */
