// k_blur.hip - 7x7 sigma=10 "Gaussian" of every pyramid level of every image of a batch in ONE launch.
//
// Semantics: reference K9 imgaussian_GPU (src/cuda/orb_gaussian.cu:21-138): for each pixel of the interior ROI
// [20,H-20) x [20,W-20): acc = 0; 49 chained single-rounding FMAs acc = fma(w[k], (float)I, acc) in raster order;
// out = trunc(acc).  Pixels outside the ROI are never written and read as 0 (the blurred
// slab is zero-filled once at create).  The weights are the hard-coded table of Appendix A.2.
// MI355X design (round 4, rewritten): the kernel is vector-ALU bound, so the design minimises vector instructions per ROI pixel and
// picks the ones that issue every 2 clocks (v_fma_f32 / v_mul_f32 / v_add_f32 run at twice the rate of v_pk_fma_f32, v_perm, v_cvt:
// profiles/r04_valu_rate.txt - a packed FMA is no faster than two plain ones, and needs its operands paired up).
//   * one LANE owns a strip of BLUR_SW = 8 output columns and walks down a band of up to BLUR_RB_MAX rows: every input row is loaded
//     (one 16-byte load, requested one row ahead) and converted ONCE (14 conversions), its 8 horizontal 7-tap sums are kept in a
//     rolling window of 7 rows in registers, and every output row is one vertical 7-tap sum of that window - 1.75 conversions and
//     14 FMAs per pixel (+ 6 rows of halo per band) where round 3's tiles (16 x 2 pixels per thread, all 8 input rows converted per
//     thread) spent 5.5 conversions and 16.6 FMAs;
//   * no LDS tile, no barrier: lanes are independent, a workgroup is just 256 consecutive (strip, band) items of one level, so small
//     levels fill their waves (round 3's 64 x 128 tiles covered 1.3x the ROI: 23 % of the lanes idle);
//   * certificate as before (below): A + 49152 rounded toward -inf leaves floor(256 A) in the mantissa; undecided pixels are flagged in
//     a per-lane byte mask in LDS, listed per wave after the band and recomputed with the reference's 49-FMA chain, one pixel per lane.
#include "jsorb_launch.h"
#include "k_blur_body.h"
#include "k_compact_body.h"

namespace jsorb {

// Work decomposition of a level (host): ROI = [20, H-20) x [20, W-20); strips of 8 columns (the first one starts BLUR_X_LEAD = 4 columns in front of
// the ROI, so that every lane's 8 output bytes are one aligned store) x bands of blur_rb rows, item = band * strips + strip
// (adjacent lanes = adjacent strips: their 16-byte loads overlap by half and stay in the L1 lines of one image row), 256 items per workgroup.
void fill_blur_layout(Geometry &g)
{
    int bblk = 0;
    for (int i = 0; i < g.L; i++) {
        LevelDesc &lv = g.lv[i];
        const int rw = lv.W - 2 * JSORB_BORDER + BLUR_X_LEAD, rh = lv.H - 2 * JSORB_BORDER;      // (BLUR_X_LEAD border columns in front of the ROI belong to the first strip)
        lv.blur_blk0 = bblk;
        if (rw <= 0 || rh <= 0) { lv.blur_bx = 1; lv.blur_by = 0; lv.blur_rb = 1; lv.blur_recip = 0; continue; }
        const int ncs = (rw + BLUR_SW - 1) / BLUR_SW;
        // single-image handles: 8-row bands - twice the workgroups, half as long (k_blur of one EuRoC image 15-17 -> 8-11 us)
        // batch handles: BLUR_RB_BATCH rows, and up to BLUR_RB_MAX on the BLUR_TALL_LEVELS largest levels (6 halo rows per band: 37 % more conversions and
        // horizontal sums at 16 rows, 19 % at 32 - but a launch of 32-row bands everywhere ended in a long thin tail: 117.2 k against 120.0 k pairs/s)
        const int rb_batch = i < BLUR_TALL_LEVELS ? BLUR_RB_TALL : BLUR_RB_BATCH;
        const int rb_max = experiment_env("JSORB_BLUR_ROWS") ? std::max(1, std::min(BLUR_RB_MAX, atoi(experiment_env("JSORB_BLUR_ROWS")))) : (g.latency ? 8 : rb_batch);
        const int nrb = (rh + rb_max - 1) / rb_max;
        lv.blur_bx = ncs;                                   // strips per band
        lv.blur_by = nrb;                                   // bands
        lv.blur_rb = (rh + nrb - 1) / nrb;                  // rows per band (the last one may be shorter)
        lv.blur_recip = ncs > 1 ? (unsigned)((0x100000000ull + ncs - 1) / ncs) : 0u;      // item / ncs = umulhi(item, recip), exact for item < 2^32 / ncs (ncs == 1: the kernel takes item itself)
        bblk += (ncs * nrb + BLUR_THREADS - 1) / BLUR_THREADS;
    }
    g.blur_blocks = bblk;
}
int blur_level_blocks(const LevelDesc &lv) { return (lv.blur_bx * lv.blur_by + BLUR_THREADS - 1) / BLUR_THREADS; }


#ifndef BLUR_MIN_WAVES
#define BLUR_MIN_WAVES 4
#endif
__global__ __launch_bounds__(BLUR_THREADS, BLUR_MIN_WAVES) void k_blur(Geometry g, ImageSrc src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *__restrict__ ctab, int n_images)
{
    // workgroup-independent arguments in the first round of scalar loads, the workgroup descriptor in the second, the level in the third
    asm volatile("" ::"s"(ctab), "s"(slab), "s"(blur_slab), "s"(src.l0), "s"(src.l0_stride), "s"(src.l0_pitch), "s"(g.slab_bytes), "s"(g.detect_blocks));
    int b, blk;
    if (!xcd_map(g.blur_blocks, n_images, b, blk)) return;
    blur_workgroup(g, src, slab, blur_slab, ctab, b, blk);
}

// Batches: k_compact and k_blur as ONE launch (round 6).  k_compact is one workgroup per image and pure latency; as a launch of its own inside the
// 4-lane pipeline it is a bubble in its lane (a handful of workgroups that wait for wave slots while the other lanes' kernels keep every CU full: the
// launch took 45-70 us there, 10 us alone, and cost the step 2.5 x its stand-alone time - tools/micro/r6_skip.py).  k_blur follows it in the lane and
// does not depend on it (it reads the pyramid, k_compact reads k_detect's tile candidates), so workgroup 0 of every image compacts (256 threads,
// 1024 / 256 x the candidates per thread) and workgroups 1 .. blur_blocks blur: the compaction runs under k_blur's workgroups.
// NC as in k_compact_flat: > 0 - the candidates stay in registers (images of at most 4096 tiles), 0 - re-reading form; tables for <= 32768 tiles.
#define BLC_MAXCELLS 512
template <int NC>
__global__ __launch_bounds__(BLUR_THREADS, BLUR_MIN_WAVES) void k_blur_compact(Geometry g, ImageSrc src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *__restrict__ ctab, int n_images,
                                                                               const unsigned long long *__restrict__ tile_out, unsigned long long *__restrict__ kp,
                                                                               int *__restrict__ counts, int *__restrict__ row_tab, int *__restrict__ counts_host)
{
    extern __shared__ int s_epi_dyn[];
    asm volatile("" ::"s"(ctab), "s"(slab), "s"(blur_slab), "s"(src.l0), "s"(src.l0_stride), "s"(src.l0_pitch), "s"(g.slab_bytes), "s"(g.detect_blocks));
    int b, blk;
    if (!xcd_map(g.blur_blocks + 1, n_images, b, blk)) return;
    if (blk == 0) {
        compact_flat_workgroup<NC, BLUR_THREADS, BLC_MAXCELLS>(g, tile_out, kp, counts, row_tab, counts_host, b, s_epi_dyn);
        return;
    }
    blur_workgroup(g, src, slab, blur_slab, ctab, b, blk - 1);
}

bool blur_compact_fusable(const Geometry &g)
{
    // tables for <= 64 * BLC_MAXCELLS tiles; the bucket counters must fit next to k_blur's static LDS in what a workgroup may request
    return g.blur_blocks > 0 && g.T <= 64 * BLC_MAXCELLS && (size_t)g.L * g.epi_rows * sizeof(int) <= 40 * 1024;
}

void launch_blur_compact(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s,
                         const unsigned long long *tile_out, unsigned long long *kp, int *counts, int *row_tab, int *counts_host)
{
    const size_t epi = g.epi_rows ? (size_t)g.L * g.epi_rows * sizeof(int) : 0;
    if (g.T <= 16 * BLUR_THREADS)
        hipLaunchKernelGGL((k_blur_compact<16>), xcd_grid(g.blur_blocks + 1, n_images), dim3(BLUR_THREADS), epi, s, g, src, slab, blur_slab, ctab, n_images, tile_out, kp, counts, row_tab, counts_host);
    else
        hipLaunchKernelGGL((k_blur_compact<0>), xcd_grid(g.blur_blocks + 1, n_images), dim3(BLUR_THREADS), epi, s, g, src, slab, blur_slab, ctab, n_images, tile_out, kp, counts, row_tab, counts_host);
}

void launch_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s)
{
    if (g.blur_blocks == 0) return;
    hipLaunchKernelGGL(k_blur, xcd_grid(g.blur_blocks, n_images), dim3(BLUR_THREADS), 0, s, g, src, slab, blur_slab, ctab, n_images);
}

} // namespace jsorb
