// k_blur.hip - 7x7 sigma=10 "Gaussian" of every pyramid level of every image of a batch in ONE launch.
//
// Semantics: reference K9 imgaussian_GPU (src/cuda/orb_gaussian.cu:21-138): for each pixel of the interior ROI
// [20,H-20) x [20,W-20): acc = 0; 49 chained single-rounding FMAs acc = fma(w[k], (float)I, acc) in raster order;
// out = trunc(acc).  Pixels outside the ROI are never written and read as 0 (the blurred
// slab is zero-filled once at create).  The weights are the hard-coded table of Appendix A.2.
// MI355X design: a 256-thread workgroup stages a (BLUR_TH+6) x (BLUR_TW+16) byte tile in LDS with 16-byte loads; each thread
// produces a 16-pixel strip of 2 adjacent rows.  The kernel is vector-ALU bound (49 FMAs + the u8 -> f32 conversions per
// pixel), so the strip is wide (22 conversions per 16 pixels per row instead of 14 per 8) and every converted input row feeds
// both output rows; the 16 independent FMA chains of a row pair up in v_pk_fma_f32 while each chain keeps the reference order.
#include "jsorb_launch.h"

namespace jsorb {

// normalised weight by squared distance d = j*j + k*k from the centre (Appendix A.2)
__host__ __device__ __forceinline__ constexpr unsigned gauss_bits(int d)
{
    return d == 0 ? 0x3CADF459u : d == 1 ? 0x3CAD163Eu : d == 2 ? 0x3CAC393Fu : d == 4 ? 0x3CAA828Du :
           d == 5 ? 0x3CA9A8D7u : d == 8 ? 0x3CA72236u : d == 9 ? 0x3CA64CD0u : d == 10 ? 0x3CA5787Bu :
           d == 13 ? 0x3CA301D1u : 0x3C9EFB81u /* d == 18 */;
}

// weight of tap (r, c) = c_gauss[r][|c - 3|]
__constant__ unsigned c_gauss_bits[7][4] = {
    {gauss_bits(9 + 0), gauss_bits(9 + 1), gauss_bits(9 + 4), gauss_bits(9 + 9)}, {gauss_bits(4 + 0), gauss_bits(4 + 1), gauss_bits(4 + 4), gauss_bits(4 + 9)},
    {gauss_bits(1 + 0), gauss_bits(1 + 1), gauss_bits(1 + 4), gauss_bits(1 + 9)}, {gauss_bits(0 + 0), gauss_bits(0 + 1), gauss_bits(0 + 4), gauss_bits(0 + 9)},
    {gauss_bits(1 + 0), gauss_bits(1 + 1), gauss_bits(1 + 4), gauss_bits(1 + 9)}, {gauss_bits(4 + 0), gauss_bits(4 + 1), gauss_bits(4 + 4), gauss_bits(4 + 9)},
    {gauss_bits(9 + 0), gauss_bits(9 + 1), gauss_bits(9 + 4), gauss_bits(9 + 9)}};
#define c_gauss reinterpret_cast<const float (*)[4]>(c_gauss_bits)

// strip of BLUR_STRIP pixels x BLUR_ROWS rows per thread; 256 threads cover a BLUR_TW x BLUR_TH tile
#ifndef BLUR_STRIP
#define BLUR_STRIP 16
#endif
#ifndef BLUR_ROWS
#define BLUR_ROWS 2
#endif
#ifndef BLUR_TW
#define BLUR_TW 64
#endif
#define BLUR_TH (256 * BLUR_ROWS * BLUR_STRIP / BLUR_TW)
#define BLUR_STRIDE (BLUR_TW + 16)
#define BLUR_HALF (BLUR_STRIP / 2)

__global__ __launch_bounds__(256) void k_blur(Geometry g, ImageSrc src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *__restrict__ ctab, int n_images)
{
    __shared__ __align__(16) unsigned char tile[(BLUR_TH + 6) * BLUR_STRIDE];
    const int tid = threadIdx.x;
    // workgroup-independent arguments in the first round of scalar loads, the workgroup descriptor (level, tile) in the second,
    // the level in the third: the first image byte cannot be requested earlier (see k_detect)
    asm volatile("" ::"s"(ctab), "s"(slab), "s"(blur_slab), "s"(src.l0), "s"(src.l0_stride), "s"(src.l0_pitch), "s"(g.slab_bytes), "s"(g.detect_blocks));
    int b, blk;
    if (!xcd_map(blockIdx.x, g.blur_blocks, n_images, b, blk)) return;
    const unsigned wd = ctab_load(ctab, ctab_blur(g) + blk);
    const int lvl = (int)(wd & 15u), by = (int)((wd >> 4) & 0x3FFFu), bx = (int)(wd >> 18);
    const LevelDesc &lv = g.lv[lvl];
    const int H = lv.H, W = lv.W;
    asm volatile("" ::"s"(lv.img_off), "s"(lv.pitch), "s"(H), "s"(W));
    const int x0 = JSORB_BORDER + bx * BLUR_TW, y0 = JSORB_BORDER + by * BLUR_TH;
    int pitch;
    const uint8_t *img = level_ptr_uniform(g, src, slab, b, lvl, lv.pitch, lv.img_off, pitch);

    // 16-byte staging loads (x0 - 4 is a multiple of 16: x0 = 20 + BLUR_TW*bx)
    constexpr int NQ = BLUR_STRIDE / 16;
    for (int i = tid; i < (BLUR_TH + 6) * NQ; i += 256) {
        const int ly = i / NQ, dx = i - ly * NQ;
        const int y = y0 - 3 + ly, x = x0 - 4 + 16 * dx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y < H && x + 16 <= pitch) v = *reinterpret_cast<const uint4 *>(img + (size_t)y * pitch + x);
        reinterpret_cast<uint4 *>(tile)[i] = v;
    }
    __syncthreads();

    constexpr int SPR = BLUR_TW / BLUR_STRIP;         // strips per tile row
    const int ty = tid / SPR, tx = tid % SPR;
    const int y = y0 + BLUR_ROWS * ty, x = x0 + BLUR_STRIP * tx;
    if (y >= H - JSORB_BORDER || x >= W - JSORB_BORDER) return;

    // BLUR_STRIP independent FMA chains per output row, evaluated two at a time with v_pk_fma_f32 (IEEE fma per component, so each
    // chain is still the reference's 49 sequential single-rounding FMAs).  A[o][j] = (acc of pixel j, acc of pixel j + HALF): for
    // tap column c its operand pair is Q[1+j+c] = (f[1+j+c], f[1+j+c+HALF]) - pairs HALF bytes apart need no re-alignment moves.
    // The kernel is vector-ALU bound and the u8 -> f32 conversions are ~40 % of it, so a thread owns a wide strip (fewer halo
    // conversions) of BLUR_ROWS adjacent rows (each converted input row feeds both output rows).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 A[BLUR_ROWS][BLUR_HALF];
#pragma unroll
    for (int o = 0; o < BLUR_ROWS; o++)
#pragma unroll
        for (int j = 0; j < BLUR_HALF; j++) A[o][j] = (f2){0.0f, 0.0f};
    // The row loop is NOT fully unrolled: unrolled, the compiler's schedule needs 140-200 VGPRs (2-3 waves per SIMD), and this
    // kernel needs the occupancy to overlap the staging phase of one workgroup with the arithmetic of the others.  The weights
    // of tap row r come from a constant table through scalar loads.  Input rows 0 and 7 feed one output row each and are peeled;
    // rows 1..6 feed both and run as 3 iterations of two rows with two ping-pong row buffers (no register copies, no branches).
    static_assert(BLUR_ROWS == 2, "the row schedule below is written for two output rows per thread");
    constexpr int NW = BLUR_STRIP / 4 + 2;
    const unsigned char *rowp = tile + (BLUR_ROWS * ty) * BLUR_STRIDE + BLUR_STRIP * tx;
    auto load_row = [&](unsigned (&w)[NW], int row) {
#pragma unroll
        for (int k = 0; k < NW / 2; k++) { const uint2 v = reinterpret_cast<const uint2 *>(rowp + row * BLUR_STRIDE)[k]; w[2 * k] = v.x; w[2 * k + 1] = v.y; }
    };
    auto convert = [&](const unsigned (&w)[NW], f2 (&Q)[BLUR_HALF + 7]) {
#pragma unroll
        for (int k = 1; k <= BLUR_HALF + 6; k++) {
            const int k2 = k + BLUR_HALF;
            Q[k] = (f2){(float)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu), (float)((w[k2 >> 2] >> (8 * (k2 & 3))) & 0xFFu)};
        }
    };
    auto accum = [&](f2 (&acc)[BLUR_HALF], const f2 (&Q)[BLUR_HALF + 7], int r) {      // r: tap row (wave-uniform)
        const float wr[4] = {c_gauss[r][0], c_gauss[r][1], c_gauss[r][2], c_gauss[r][3]};
#pragma unroll
        for (int c = 0; c < 7; c++) {
            const float w = wr[c < 3 ? 3 - c : c - 3];
            const f2 w2 = (f2){w, w};
#pragma unroll
            for (int j = 0; j < BLUR_HALF; j++) acc[j] = __builtin_elementwise_fma(w2, Q[1 + j + c], acc[j]);
        }
    };
    unsigned wa[NW], wb[NW];
    f2 Q[BLUR_HALF + 7];
    load_row(wa, 0);
    load_row(wb, 1);
    convert(wa, Q); accum(A[0], Q, 0);                 // input row 0: tap row 0 of output row 0
    load_row(wa, 2);
#pragma unroll 1
    for (int i = 1; i < 7; i += 2) {                   // input rows i (in wb) and i + 1 (in wa)
        convert(wb, Q); accum(A[0], Q, i); accum(A[1], Q, i - 1);
        load_row(wb, i + 2);                           // rows 3, 5, 7
        convert(wa, Q); accum(A[0], Q, i + 1); accum(A[1], Q, i);
        load_row(wa, i + 3 < 8 ? i + 3 : 7);           // rows 4, 6 (the last load is unused)
    }
    convert(wb, Q); accum(A[1], Q, 6);                 // input row 7: tap row 6 of output row 1
    const int n_valid = (W - JSORB_BORDER) - x;      // pixels of the strip inside the ROI
#pragma unroll
    for (int o = 0; o < BLUR_ROWS; o++) {
        if (y + o >= H - JSORB_BORDER) break;
        unsigned ow[BLUR_STRIP / 4];
#pragma unroll
        for (int k = 0; k < BLUR_STRIP / 4; k++) ow[k] = 0;
#pragma unroll
        for (int j = 0; j < BLUR_HALF; j++) {
            ow[j >> 2] |= ((unsigned)A[o][j].x & 0xFFu) << (8 * (j & 3));
            ow[(j + BLUR_HALF) >> 2] |= ((unsigned)A[o][j].y & 0xFFu) << (8 * ((j + BLUR_HALF) & 3));
        }
        uint8_t *dst = blur_slab + (size_t)b * g.slab_bytes + lv.img_off + (size_t)(y + o) * lv.pitch + x;
        if (n_valid >= BLUR_STRIP) {
#pragma unroll
            for (int k = 0; k < BLUR_STRIP / 4; k++) reinterpret_cast<unsigned *>(dst)[k] = ow[k];
        } else {
#pragma unroll
            for (int j = 0; j < BLUR_STRIP; j++)
                if (j < n_valid) dst[j] = (uint8_t)((ow[j >> 2] >> (8 * (j & 3))) & 0xFFu);
        }
    }
}

void blur_tile_dims(int *tw, int *th) { *tw = BLUR_TW; *th = BLUR_TH; }

void launch_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s)
{
    if (g.blur_blocks == 0) return;
    hipLaunchKernelGGL(k_blur, dim3(xcd_grid(g.blur_blocks, n_images)), dim3(256), 0, s, g, src, slab, blur_slab, ctab, n_images);
}

} // namespace jsorb
