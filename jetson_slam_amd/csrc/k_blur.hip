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

// ---- certified fast path -------------------------------------------------------------------------------------------------
// The reference's value is C = trunc(chain), the chain being 49 sequentially rounded FMAs.  The weights are (up to float rounding)
// an outer product w[j][k] ~ gv[j] * gh[k], so the same real-valued sum S can be approximated by a separable evaluation A
// (7 vertical + 7 horizontal FMAs per pixel instead of 49).  Both C and A are within rigorous bounds of S:
//   |C - S| <= gamma_49 * 255 * sum(w)                          = 7.45e-4      (gamma_n = n u / (1 - n u), u = 2^-24)
//   |A - S| <= rounding of the two 7-FMA stages + 255 * sum |gv[j] gh[k] - w[j][k]|  = 2.13e-4 + 0.9e-5
// (tests/test_blur_certificate.py recomputes both from the tables with exact rational arithmetic), hence |A - C| <= 9.7e-4.
// A pixel whose A is farther than BLUR_BAND = 2^-8 = 3.9e-3 from an integer therefore has floor(C) = floor(A) - decided with ONE
// magic-number addition rounded down (round 3; two roundings, two clamps and two subtractions per pixel pair before): floor(256 A)
// lands in the mantissa, its high byte is the result and a low byte of 0 or 255 marks the pixel as undecided.  Those (~0.8 % of
// natural pixels; every pixel of an exactly flat window, whose C lies within 1e-4 of an integer) are listed per workgroup and
// recomputed with the exact chain; a tile with too many of them is recomputed densely by the exact strip code.  The output is
// bit-identical to the chain in every case.
#define BLUR_BAND 0.00390625f
#define BLUR_AMB_CAP 1024      // listed ambiguous pixels per workgroup (of 8192) before the dense exact path takes over

// separable factors: gv[j] = exp(-j^2/200) and gh[k] = exp(-k^2/200) / 47.092777252197266 (the reference's f32 weight sum 0x423C5F01),
// rounded to f32 from double; sum |gv[j] gh[k] - w[j][k]| = 3.4e-8 for these
__constant__ float c_sep_v[4] = {1.0f, 0.99501247919268232f, 0.98019867330675525f, 0.95599748183309996f};
__constant__ float c_sep_h[4] = {(float)(1.0 / 47.092777252197266), (float)(0.99501247919268232 / 47.092777252197266),
                                 (float)(0.98019867330675525 / 47.092777252197266), (float)(0.95599748183309996 / 47.092777252197266)};

__global__ __launch_bounds__(256) void k_blur(Geometry g, ImageSrc src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *__restrict__ ctab, int n_images)
{
    __shared__ __align__(16) unsigned char tile[(BLUR_TH + 6) * BLUR_STRIDE];
    __shared__ unsigned short s_amb[BLUR_AMB_CAP];
    __shared__ int s_namb;
    const int tid = threadIdx.x;
    // workgroup-independent arguments in the first round of scalar loads, the workgroup descriptor (level, tile) in the second,
    // the level in the third: the first image byte cannot be requested earlier (see k_detect)
    asm volatile("" ::"s"(ctab), "s"(slab), "s"(blur_slab), "s"(src.l0), "s"(src.l0_stride), "s"(src.l0_pitch), "s"(g.slab_bytes), "s"(g.detect_blocks));
    int b, blk;
    if (!xcd_map(g.blur_blocks, n_images, b, blk)) return;
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
    if (tid == 0) s_namb = 0;
    __syncthreads();

    constexpr int SPR = BLUR_TW / BLUR_STRIP;         // strips per tile row
    const int ty = tid / SPR, tx = tid % SPR;
    const int y = y0 + BLUR_ROWS * ty, x = x0 + BLUR_STRIP * tx;
    const bool active = y < H - JSORB_BORDER && x < W - JSORB_BORDER;      // no early return: two more workgroup barriers follow
    const int n_valid = (W - JSORB_BORDER) - x;      // pixels of the strip inside the ROI
    uint8_t *const out_base = blur_slab + (size_t)b * g.slab_bytes + lv.img_off;

    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(BLUR_ROWS == 2 && BLUR_STRIP == 16, "the schedules below are written for 16 px x 2 rows per thread");
    constexpr int NW = BLUR_STRIP / 4 + 2;
    constexpr int NP = BLUR_HALF + 6;                 // 14 operand pairs (f[k], f[k + 8]), k = 1 .. 14: window columns of the 16 pixels
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
    auto store_rows = [&](const unsigned (&ow)[BLUR_ROWS][BLUR_STRIP / 4]) {
#pragma unroll
        for (int o = 0; o < BLUR_ROWS; o++) {
            if (y + o >= H - JSORB_BORDER) break;
            uint8_t *dst = out_base + (size_t)(y + o) * lv.pitch + x;
            if (n_valid >= BLUR_STRIP) {
#pragma unroll
                for (int k = 0; k < BLUR_STRIP / 4; k++) reinterpret_cast<unsigned *>(dst)[k] = ow[o][k];
            } else {
#pragma unroll
                for (int j = 0; j < BLUR_STRIP; j++)
                    if (j < n_valid) dst[j] = (uint8_t)((ow[o][j >> 2] >> (8 * (j & 3))) & 0xFFu);
            }
        }
    };

    // ---- fast pass: separable evaluation + certificate ----
    if (active) {
        // The horizontal stage wants its operands as pairs of window columns 8 apart, (k, k + 8) for k = 1 .. 14: 28 column values for
        // the 22 distinct columns of the strip.  The vertical stage therefore runs on a COMPACT set of 11 pairs that holds every column
        // once - (1,2) (3,4) (5,6) and (7,15) (8,16) .. (14,22) - and the six pairs (k, k + 8), k = 1 .. 6, are put together from it
        // afterwards (one move per pair and output row): 22 conversions and 11 packed FMAs per input row and output row instead of 28 / 14.
        constexpr int NC = 11;
        f2 CV[BLUR_ROWS][NC];
        unsigned wa[NW];
        f2 CQ[NC];
        auto convert_compact = [&](const unsigned (&w)[NW], f2 (&Q)[NC]) {
            auto byte_f = [&](int k) { return (float)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu); };
#pragma unroll
            for (int m = 0; m < 3; m++) Q[m] = (f2){byte_f(1 + 2 * m), byte_f(2 + 2 * m)};
#pragma unroll
            for (int m = 0; m < 8; m++) Q[3 + m] = (f2){byte_f(7 + m), byte_f(15 + m)};
        };
        // vertical stage: input row r feeds tap row r of output row 0 (r <= 6) and tap row r - 1 of output row 1 (r >= 1); rows 0 and 7
        // feed one output row each and are peeled, rows 1 .. 6 run as a rolled loop (weights through scalar loads)
        load_row(wa, 0);
        convert_compact(wa, CQ);
        {
            const f2 g02 = (f2){c_sep_v[3], c_sep_v[3]};
#pragma unroll
            for (int k = 0; k < NC; k++) { CV[0][k] = g02 * CQ[k]; CV[1][k] = (f2){0.0f, 0.0f}; }
        }
#pragma unroll 1
        for (int r = 1; r < 7; r++) {
            load_row(wa, r);
            convert_compact(wa, CQ);
            const int t0 = r < 4 ? 3 - r : r - 3, t1 = r < 5 ? 4 - r : r - 4;      // |tap row - 3| (wave-uniform)
            const float g0 = c_sep_v[t0], g1 = c_sep_v[t1];
            const f2 g02 = (f2){g0, g0}, g12 = (f2){g1, g1};
#pragma unroll
            for (int k = 0; k < NC; k++) {
                CV[0][k] = __builtin_elementwise_fma(g02, CQ[k], CV[0][k]);
                CV[1][k] = __builtin_elementwise_fma(g12, CQ[k], CV[1][k]);
            }
        }
        load_row(wa, 7);
        convert_compact(wa, CQ);
        {
            const f2 g12 = (f2){c_sep_v[3], c_sep_v[3]};
#pragma unroll
            for (int k = 0; k < NC; k++) CV[1][k] = __builtin_elementwise_fma(g12, CQ[k], CV[1][k]);
        }
        // the operand pairs of the horizontal stage: VA[o][k'] = columns (k' + 1, k' + 9)
        f2 VA[BLUR_ROWS][NP];
#pragma unroll
        for (int o = 0; o < BLUR_ROWS; o++) {
#pragma unroll
            for (int k = 0; k < 6; k++) VA[o][k] = (f2){(k & 1) ? CV[o][k >> 1].y : CV[o][k >> 1].x, CV[o][k + 5].x};
#pragma unroll
            for (int k = 6; k < NP; k++) VA[o][k] = CV[o][k - 3];
        }
        // horizontal stage + certificate, pixel pair (j, j + 8) at a time
        const f2 gh2[4] = {(f2){c_sep_h[0], c_sep_h[0]}, (f2){c_sep_h[1], c_sep_h[1]}, (f2){c_sep_h[2], c_sep_h[2]}, (f2){c_sep_h[3], c_sep_h[3]}};
        const f2 magic = (f2){49152.0f, 49152.0f};
        unsigned ow[BLUR_ROWS][BLUR_STRIP / 4];
        unsigned amb = 0;                              // undecided pixels: byte t, bit 7 - k  <=>  pixel 4 (k & 3) + t of output row k >> 2
#pragma unroll
        for (int o = 0; o < BLUR_ROWS; o++) {
            f2 h[BLUR_HALF];
#pragma unroll
            for (int j = 0; j < BLUR_HALF; j++) {
                f2 t = gh2[3] * VA[o][j];
                t = __builtin_elementwise_fma(gh2[2], VA[o][j + 1], t);
                t = __builtin_elementwise_fma(gh2[1], VA[o][j + 2], t);
                t = __builtin_elementwise_fma(gh2[0], VA[o][j + 3], t);
                t = __builtin_elementwise_fma(gh2[1], VA[o][j + 4], t);
                t = __builtin_elementwise_fma(gh2[2], VA[o][j + 5], t);
                h[j] = __builtin_elementwise_fma(gh2[3], VA[o][j + 6], t);
            }
            // q = floor(256 A) in the low 16 mantissa bits of A + 49152 (ulp 2^-8) ROUNDED DOWN: the eight additions run with the wave's f32
            // rounding mode switched to -inf (everything else in this kernel is round-to-nearest-even).  Bits 8-15 are floor(A); a
            // fraction byte of 0 or 255 puts A within 2^-8 = 3.9e-3 of an integer, four times the bound on |A - C|: undecided.
            f2 r[BLUR_HALF];
            static_assert(BLUR_HALF == 8, "the asm statement below adds eight pixel pairs");
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 1\n\t"
                         "v_pk_add_f32 %0, %8, %16\n\tv_pk_add_f32 %1, %9, %16\n\tv_pk_add_f32 %2, %10, %16\n\tv_pk_add_f32 %3, %11, %16\n\t"
                         "v_pk_add_f32 %4, %12, %16\n\tv_pk_add_f32 %5, %13, %16\n\tv_pk_add_f32 %6, %14, %16\n\tv_pk_add_f32 %7, %15, %16\n\t"
                         "s_nop 1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                         : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]), "v"(magic));
#pragma unroll
            for (int wi = 0; wi < BLUR_STRIP / 4; wi++) {
                // pixels 4 wi .. 4 wi + 3 of the strip: component wi >> 1 of the pairs 4 (wi & 1) .. 4 (wi & 1) + 3
                unsigned q4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) q4[t] = __float_as_uint((wi >> 1) ? r[4 * (wi & 1) + t].y : r[4 * (wi & 1) + t].x);
                const unsigned a01 = __builtin_amdgcn_perm(q4[1], q4[0], 0x04000501u), a23 = __builtin_amdgcn_perm(q4[3], q4[2], 0x04000501u);
                ow[o][wi] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);               // floor(A) of the four pixels
                const unsigned fr = __builtin_amdgcn_perm(a23, a01, 0x07060302u);       // floor(256 A) mod 256
                // y = (f ^ f << 1) & 0xFE is zero exactly for fraction bytes 0 and 255; ~y & (y - 0x01010101) has bit 7 of every zero byte of
                // y set (and possibly that of a byte of value 1 above one, which only lists a decided pixel as well)
                const unsigned y = (fr ^ (fr << 1)) & 0xFEFEFEFEu;
                const unsigned z = ~y & (y - 0x01010101u);
                constexpr int k = 0;
                (void)k;
                amb |= (z >> (4 * o + wi)) & (0x80808080u >> (4 * o + wi));
            }
        }
        store_rows(ow);
        // list the pixels that need the exact chain (inside the ROI only)
        while (amb) {
            const int pos = __builtin_ctz(amb);
            amb &= amb - 1;
            const int k = 7 - (pos & 7), o = k >> 2, p = 4 * (k & 3) + (pos >> 3);
            if (p < n_valid && y + o < H - JSORB_BORDER) {
                const int idx = atomicAdd(&s_namb, 1);
                if (idx < BLUR_AMB_CAP) s_amb[idx] = (unsigned short)((BLUR_ROWS * ty + o) * BLUR_TW + BLUR_STRIP * tx + p);
            }
        }
    }
    __syncthreads();      // also orders the dword stores above before the byte stores below (same addresses, other threads)
    const int n_amb = s_namb;
    if (n_amb == 0) return;
    if (n_amb <= BLUR_AMB_CAP) {
        // ---- exact chain for the listed pixels, one per lane: acc = fma(w[r][c], I, acc) in raster order ----
        for (int i = tid; i < n_amb; i += 256) {
            const int id = s_amb[i], ly = id / BLUR_TW, lx = id - ly * BLUR_TW;
            const unsigned char *wp = tile + ly * BLUR_STRIDE + lx + 1;      // window row 0, column 0 (tile column 0 is x0 - 4)
            float acc = 0.0f;
#pragma unroll 1
            for (int r = 0; r < 7; r++) {
                const float w0 = c_gauss[r][0], w1 = c_gauss[r][1], w2 = c_gauss[r][2], w3 = c_gauss[r][3];
                const unsigned char *q = wp + r * BLUR_STRIDE;
                acc = __builtin_fmaf(w3, (float)q[0], acc);
                acc = __builtin_fmaf(w2, (float)q[1], acc);
                acc = __builtin_fmaf(w1, (float)q[2], acc);
                acc = __builtin_fmaf(w0, (float)q[3], acc);
                acc = __builtin_fmaf(w1, (float)q[4], acc);
                acc = __builtin_fmaf(w2, (float)q[5], acc);
                acc = __builtin_fmaf(w3, (float)q[6], acc);
            }
            out_base[(size_t)(y0 + ly) * lv.pitch + x0 + lx] = (uint8_t)((unsigned)acc & 0xFFu);
        }
        return;
    }
    if (!active) return;

    // ---- dense exact path (a tile of mostly flat windows): the 49-FMA chains of the whole strip, two at a time ----
    // BLUR_STRIP independent FMA chains per output row, evaluated two at a time with v_pk_fma_f32 (IEEE fma per component, so each
    // chain is still the reference's 49 sequential single-rounding FMAs).  A[o][j] = (acc of pixel j, acc of pixel j + HALF): for
    // tap column c its operand pair is Q[1+j+c] = (f[1+j+c], f[1+j+c+HALF]) - pairs HALF bytes apart need no re-alignment moves.
    f2 A[BLUR_ROWS][BLUR_HALF];
#pragma unroll
    for (int o = 0; o < BLUR_ROWS; o++)
#pragma unroll
        for (int j = 0; j < BLUR_HALF; j++) A[o][j] = (f2){0.0f, 0.0f};
    // The row loop is NOT fully unrolled: unrolled, the compiler's schedule needs 140-200 VGPRs.  The weights of tap row r come from a
    // constant table through scalar loads.  Input rows 0 and 7 feed one output row each and are peeled; rows 1..6 feed both and run
    // as 3 iterations of two rows with two ping-pong row buffers (no register copies, no branches).
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
    unsigned ow[BLUR_ROWS][BLUR_STRIP / 4];
#pragma unroll
    for (int o = 0; o < BLUR_ROWS; o++) {
#pragma unroll
        for (int k = 0; k < BLUR_STRIP / 4; k++) ow[o][k] = 0;
#pragma unroll
        for (int j = 0; j < BLUR_HALF; j++) {
            ow[o][j >> 2] |= ((unsigned)A[o][j].x & 0xFFu) << (8 * (j & 3));
            ow[o][(j + BLUR_HALF) >> 2] |= ((unsigned)A[o][j].y & 0xFFu) << (8 * ((j + BLUR_HALF) & 3));
        }
    }
    store_rows(ow);
}

void blur_tile_dims(int *tw, int *th) { *tw = BLUR_TW; *th = BLUR_TH; }

void launch_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *ctab, int n_images, hipStream_t s)
{
    if (g.blur_blocks == 0) return;
    hipLaunchKernelGGL(k_blur, xcd_grid(g.blur_blocks, n_images), dim3(256), 0, s, g, src, slab, blur_slab, ctab, n_images);
}

} // namespace jsorb
