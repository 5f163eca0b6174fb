// k_blur.hip - 7x7 sigma=10 "Gaussian" of every pyramid level of every image of a batch in ONE launch.
//
// Semantics: reference K9 imgaussian_GPU (src/cuda/orb_gaussian.cu:21-138): for each pixel of the interior ROI
// [20,H-20) x [20,W-20): acc = 0; 49 chained single-rounding FMAs acc = fma(w[k], (float)I, acc) in raster order;
// out = trunc(acc).  Pixels outside the ROI are never written and read as 0 (the blurred
// slab is zero-filled once at create).  The weights are the hard-coded table of Appendix A.2.
// MI355X design: a 256-thread workgroup stages a (32+6) x 80 byte tile in LDS with 190 16-byte loads; each
// thread produces an 8-pixel strip, reading every tile row as two ds_read_b64 and converting each byte once per
// row (14 v_cvt_f32_ubyte instead of 56); the 8 independent FMA chains interleave freely while each chain keeps
// the reference order.
#include "jsorb_launch.h"

namespace jsorb {

// normalised weight by squared distance d = j*j + k*k from the centre (Appendix A.2)
__device__ __forceinline__ constexpr unsigned gauss_bits(int d)
{
    return d == 0 ? 0x3CADF459u : d == 1 ? 0x3CAD163Eu : d == 2 ? 0x3CAC393Fu : d == 4 ? 0x3CAA828Du :
           d == 5 ? 0x3CA9A8D7u : d == 8 ? 0x3CA72236u : d == 9 ? 0x3CA64CD0u : d == 10 ? 0x3CA5787Bu :
           d == 13 ? 0x3CA301D1u : 0x3C9EFB81u /* d == 18 */;
}

#define BLUR_TW 64
#define BLUR_TH 32
#define BLUR_STRIDE 80

__global__ __launch_bounds__(256) void k_blur(Geometry g, ImageSrc src, const uint8_t *slab, uint8_t *blur_slab, int n_images)
{
    __shared__ __align__(16) unsigned char tile[(BLUR_TH + 6) * BLUR_STRIDE];
    const int tid = threadIdx.x;
    int b, blk;
    if (!xcd_map(blockIdx.x, g.blur_blocks, n_images, b, blk)) return;
    int lvl = 0;
#pragma unroll 1
    for (int i = 1; i < g.L; i++)
        if (blk >= g.lv[i].blur_blk0) lvl = i;
    const LevelDesc &lv = g.lv[lvl];
    const int lb = blk - lv.blur_blk0;
    const int bx = lb % lv.blur_bx, by = lb / lv.blur_bx;
    const int H = lv.H, W = lv.W;
    const int x0 = JSORB_BORDER + bx * BLUR_TW, y0 = JSORB_BORDER + by * BLUR_TH;
    int pitch;
    const uint8_t *img = level_ptr(g, src, slab, b, lvl, pitch);

    // 16-byte staging loads (x0 - 4 is a multiple of 16: x0 = 20 + 64*bx); a tile row is 5 x 16 B
    if (tid < (BLUR_TH + 6) * (BLUR_STRIDE / 16)) {
        const int ly = tid / (BLUR_STRIDE / 16), dx = tid - ly * (BLUR_STRIDE / 16);
        const int y = y0 - 3 + ly, x = x0 - 4 + 16 * dx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (y < H && x + 16 <= pitch) v = *reinterpret_cast<const uint4 *>(img + (size_t)y * pitch + x);
        reinterpret_cast<uint4 *>(tile)[tid] = v;
    }
    __syncthreads();

    const int ty = tid >> 3, tx = tid & 7;
    const int y = y0 + ty, x = x0 + 8 * tx;
    if (y >= H - JSORB_BORDER || x >= W - JSORB_BORDER) return;

    // 8 independent FMA chains, evaluated two at a time with v_pk_fma_f32 (IEEE fma per component, so each chain is still the
    // reference's 49 sequential single-rounding FMAs).  A[j] = (acc of pixel j, acc of pixel j+4): for tap column c its operand
    // pair is Q[1+j+c] = (f[1+j+c], f[5+j+c]) - pairs four bytes apart need no re-alignment moves, only 6 duplicate conversions.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 A[4];
#pragma unroll
    for (int j = 0; j < 4; j++) A[j] = (f2){0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const uint2 *p = reinterpret_cast<const uint2 *>(tile + (ty + r) * BLUR_STRIDE + 8 * tx);
        const uint2 lo = p[0], hi = p[1];
        const unsigned w4[4] = {lo.x, lo.y, hi.x, hi.y};
        f2 Q[11];
#pragma unroll
        for (int k = 1; k <= 10; k++) {
            const int k2 = k + 4;
            Q[k] = (f2){(float)((w4[k >> 2] >> (8 * (k & 3))) & 0xFFu), (float)((w4[k2 >> 2] >> (8 * (k2 & 3))) & 0xFFu)};
        }
#pragma unroll
        for (int c = 0; c < 7; c++) {
            const float w = __uint_as_float(gauss_bits((r - 3) * (r - 3) + (c - 3) * (c - 3)));
            const f2 w2 = (f2){w, w};
#pragma unroll
            for (int j = 0; j < 4; j++) A[j] = __builtin_elementwise_fma(w2, Q[1 + j + c], A[j]);
        }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 4; j++) { acc[j] = A[j].x; acc[j + 4] = A[j].y; }
    unsigned o0 = 0, o1 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        o0 |= ((unsigned)acc[j] & 0xFFu) << (8 * j);
        o1 |= ((unsigned)acc[4 + j] & 0xFFu) << (8 * j);
    }
    uint8_t *dst = blur_slab + (size_t)b * g.slab_bytes + lv.img_off + (size_t)y * lv.pitch + x;
    const int n_valid = (W - JSORB_BORDER) - x;      // pixels of the strip inside the ROI
    if (n_valid >= 8) {
        reinterpret_cast<unsigned *>(dst)[0] = o0;
        reinterpret_cast<unsigned *>(dst)[1] = o1;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (j < n_valid) dst[j] = (uint8_t)(((j < 4 ? o0 : o1) >> (8 * (j & 3))) & 0xFFu);
    }
}

void launch_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, int n_images, hipStream_t s)
{
    if (g.blur_blocks == 0) return;
    hipLaunchKernelGGL(k_blur, dim3(xcd_grid(g.blur_blocks, n_images)), dim3(256), 0, s, g, src, slab, blur_slab, n_images);
}

} // namespace jsorb
