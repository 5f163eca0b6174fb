// jsorb_launch.h - host-callable launchers of the gfx950 kernels (one translation unit per stage).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/jsorb.h"
#include "jsorb_device.h"
#include "jsorb_env.h"

namespace jsorb {

struct StereoArgs {
    float maxD;        // mbf / mb                 (orb_stereo_match.cu:146)
    float mbf;
    int th_high;       // ORBmatcher::TH_HIGH
    int th_orb;        // (TH_HIGH + TH_LOW) / 2    (orb_stereo_match.cu:226)
};

// dynamic LDS bytes the detect kernel needs for the given geometry (max over levels)
size_t detect_lds_bytes(const Geometry &g);

size_t pyramid_lds_bytes(const Geometry &g);
int pyramid_ns_dispatched(int ns16);      // the NS template argument k_pyramid runs for a level that needs ns16 loads per row (1, 2 or 4)
int pyramid_loads_per_row(float s, int W); // 16-byte loads per lane and level-0 row in k_pyramid for a level of scale s and width W (exact, by enumeration)
int detect_swar6_threshold(int threshold);   // k_detect: threshold of the 6-bit early rejects if they provably accept a superset of the exact ones (exhaustive check), else 0
void fill_pyramid_layout(Geometry &g);     // rows per k_pyramid tile (pyr_th), sparse windows, workgroup table offsets (host side, once per handle)
void launch_upload_level0(const uint8_t *host_pinned, uint8_t *dst, size_t bytes, hipStream_t s);      // bytes: a multiple of 16
void launch_copy_level0(const uint8_t *src, size_t image_stride, int step, uint8_t *slab, size_t slab_bytes, int pitch, int W, int H, int n_images, hipStream_t s);
void launch_pyramid(const Geometry &g, const ImageSrc &src, uint8_t *slab, const uint32_t *ctab, int n_images, size_t lds_bytes, hipStream_t s);
int detect_ring_bit_of_pixel(int k);       // bit of ring pixel k in the index of the arc LUT as k_detect forms it (the host stores the LUT in that order)
// images with up to this many tiles: k_compact as a launch of its own runs its re-reading form with 256-thread workgroups on batch handles (k_compact.hip), and a
// batch of one lane takes the fused k_blur_compact launch (jsorb_api.hip: run_pipeline)
#ifndef CMP_MID_T
#define CMP_MID_T 8192
#endif
void fill_detect_layout(Geometry &g);      // tile rows per workgroup (det_R), workgroup table offsets and per-level LDS layout of k_detect (host side, once per handle)
void launch_detect(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *mask_slab,
                   const uint32_t *lut_bits, unsigned long long *tile_out, int n_images, size_t lds_bytes, hipStream_t s,
                   unsigned *spill = nullptr, unsigned *spill_flags = nullptr);      // compact handles: the arena of spill chunks and its busy flags
int detect_spill_chunk_entries(const Geometry &g);      // u32 entries of one spill chunk (the handle's largest band region)
size_t detect_arena_bytes(const Geometry &g);           // the whole arena: 8 XCDs x slots x chunk
size_t detect_arena_flag_words();                       // one busy flag per chunk
bool detect_arena_covers(int compute_units);           // the arena has a chunk for every workgroup of the compact k_detect that can be resident on a device of this size (8 XCDs of <= 32 CUs)
int detect_pos_cap(const Geometry &g, int level);   // entries of a k_detect workgroup's pool of positives on that level (compact form)
void launch_nms_ms(const Geometry &g, unsigned long long *tile_out, int *ms_grid, int *ms_scratch, int mode_gpu, int n_images, hipStream_t s);
void launch_compact(const Geometry &g, const unsigned long long *tile_out, unsigned long long *kp, int *counts,
                    int *row_tab, int n_images, hipStream_t s, int *counts_host = nullptr);
void launch_detect_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *mask_slab, const uint32_t *lut_bits,
                        unsigned long long *tile_out, uint8_t *blur_slab, size_t lds_bytes, hipStream_t s);      // single image: k_detect and k_blur as one launch
void fill_blur_layout(Geometry &g);        // k_blur: strips x bands per level, workgroups per level (host side, once per handle)
int blur_level_blocks(const LevelDesc &lv);
void launch_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s);
bool blur_compact_fusable(const Geometry &g);      // batches: k_compact as workgroup 0 of every image of the k_blur launch (k_blur.hip)
void launch_blur_compact(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s,
                         const unsigned long long *tile_out, unsigned long long *kp, int *counts, int *row_tab, int *counts_host);
void launch_describe(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *blur_slab,
                     const unsigned long long *kp, const int *counts, float *angles, uint8_t *desc, int32_t *out_kp,
                     int n_images, hipStream_t s, Deliver dl = Deliver{nullptr, nullptr, nullptr, nullptr, nullptr});
const void *describe_kernel_address();    // host-side address of k_describe (identifies its node in a captured graph)
int describe_kernel_deliver_arg();         // index of its `Deliver dl` argument
void launch_stereo(const Geometry &g, const ImageSrc &srcL, const uint8_t *slabL, const ImageSrc &srcR, const uint8_t *slabR,
                   const int32_t *outL, const int *countsL, const uint8_t *descL,
                   const int32_t *outR, const int *countsR, const uint8_t *descR, const int *row_tabR,
                   float *u_right, float *depth, int *best_l1, unsigned *aux, StereoArgs a, int n_pairs, hipStream_t s, int *diag = nullptr);
#define JSORB_STEREO_DIAG_INTS 13          // per left keypoint: best right index, its Hamming distance, 11 L1 window sums (jsorb_copy_stereo_diagnostics)
void launch_unpack_keypoints(const int32_t *soa, int n, jsorb_keypoint *out, hipStream_t s);
void launch_assign_grid(const int32_t *soa, int n, float min_x, float min_y, float inv_w, float inv_h, int cols, int rows,
                        int32_t *cell_start, int32_t *cell_items, hipStream_t s);
void launch_gather_counts(const int *countsL, const int *countsR, const int *stats, int32_t *dst, int n_pairs, hipStream_t s);
void launch_median(const Geometry &g, const int *countsL, float *u_right, float *depth, const int *best_l1, const unsigned *aux,
                   int *stats, int n_pairs, hipStream_t s, DeliverStereo dl = DeliverStereo{nullptr, nullptr, nullptr});

} // namespace jsorb
