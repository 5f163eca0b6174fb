// k_blur_body.h - device side of k_blur (see k_blur.hip for the design): the work of ONE workgroup as a function, so that it can be a kernel of its
// own (k_blur, batches) and one half of the fused k_detect_blur launch of single frames (k_detect.hip).
#pragma once
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
static __constant__ unsigned c_gauss_bits[7][4] = {
    {gauss_bits(9 + 0), gauss_bits(9 + 1), gauss_bits(9 + 4), gauss_bits(9 + 9)}, {gauss_bits(4 + 0), gauss_bits(4 + 1), gauss_bits(4 + 4), gauss_bits(4 + 9)},
    {gauss_bits(1 + 0), gauss_bits(1 + 1), gauss_bits(1 + 4), gauss_bits(1 + 9)}, {gauss_bits(0 + 0), gauss_bits(0 + 1), gauss_bits(0 + 4), gauss_bits(0 + 9)},
    {gauss_bits(1 + 0), gauss_bits(1 + 1), gauss_bits(1 + 4), gauss_bits(1 + 9)}, {gauss_bits(4 + 0), gauss_bits(4 + 1), gauss_bits(4 + 4), gauss_bits(4 + 9)},
    {gauss_bits(9 + 0), gauss_bits(9 + 1), gauss_bits(9 + 4), gauss_bits(9 + 9)}};
#define c_gauss reinterpret_cast<const float (*)[4]>(c_gauss_bits)

#define BLUR_SW 8              // output pixels per lane and row (round 6 measured 4 - 76 instead of 105 VGPRs, 6 instead of 4 waves per SIMD, twice the lanes:
                               // k_blur 0.191 against 0.185 ms per step, the pipeline 2 % slower; profiles/r06_experiments.txt)
#ifndef BLUR_RB_MAX
#define BLUR_RB_MAX 16         // most output rows per lane; per level the host evens the bands out (fill_blur_layout).  Measured at C2 (pairs/s): 32 rows
                               // 117.2 k, 16 rows 120.0 k - six halo rows per band cost 37 % more conversions and horizontal sums, but a wave
                               // of 32-row bands lives for a third of the whole launch and the launch ends in a long, thin tail
#endif
#ifndef BLUR_RB_BATCH
#define BLUR_RB_BATCH 16       // output rows per lane on the levels that are not "tall" (batch handles)
#endif
#ifndef BLUR_TALL_LEVELS
#define BLUR_TALL_LEVELS 0     // the first n levels of a batch handle take bands of BLUR_RB_TALL rows
#endif
#ifndef BLUR_RB_TALL
#define BLUR_RB_TALL BLUR_RB_MAX
#endif
static_assert(BLUR_RB_TALL <= BLUR_RB_MAX && BLUR_RB_BATCH <= BLUR_RB_MAX, "band heights are bounded by the per-lane mask bytes");
// Columns of the image border in front of the ROI that the FIRST strip of a row covers (0 or 4).  With 4, a lane's 8 output columns start at
// 16 + 8 * strip: its store is one aligned 8-byte store (at 20 + 8 * strip it is two dword stores that straddle 8-byte units), and the four border
// columns it covers are written as 0 - what the blurred plane holds outside the ROI anyway.
// (measured at C2, A/B on one box: k_blur 0.192 -> 0.189 ms per step, 120.8 k -> 121.5 k pairs/s; 32-row bands on the 1 / 3 largest levels, tried in the same
// run: k_blur 0.209 ms, 119.4 k / 118.4 k - the tail again)
#ifndef BLUR_X_LEAD
#define BLUR_X_LEAD 4
#endif
static_assert(BLUR_RB_MAX % 16 == 0 && BLUR_RB_MAX <= 32, "the per-lane row masks are read back as 16-byte units; list entries hold 5 bits of row");
#ifndef BLUR_BOUSTRO
#define BLUR_BOUSTRO 1         // odd bands walk bottom-up (see blur_workgroup)
#endif
#ifndef BLUR_PREFETCH
#define BLUR_PREFETCH 2        // input rows requested ahead of the one being evaluated
#endif
#define BLUR_THREADS 256
#define BLUR_AMB_CAP 768       // listed undecided pixels per WAVE (of <= 64 * 8 * BLUR_RB_MAX = 16384) before the dense exact path takes over

// ---- certified fast path -------------------------------------------------------------------------------------------------
// The reference's value is C = trunc(chain), the chain being 49 sequentially rounded FMAs.  The weights are (up to float rounding)
// an outer product w[j][k] ~ gv[j] * gh[k], so the same real-valued sum S can be approximated by a separable evaluation A
// (7 horizontal + 7 vertical FMAs per pixel instead of 49).  Both C and A are within rigorous bounds of S:
//   |C - S| <= gamma_49 * 255 * sum(w)                          = 7.45e-4      (gamma_n = n u / (1 - n u), u = 2^-24)
//   |A - S| <= rounding of the two 7-FMA stages + 255 * sum |gv[j] gh[k] - w[j][k]|  = 2.1e-4 + 0.9e-5
// (tests/test_blur_certificate.py recomputes both from the tables with exact rational arithmetic), hence |A - C| <= 9.7e-4.
// A pixel whose A is farther than BLUR_BAND = 2^-8 = 3.9e-3 from an integer therefore has floor(C) = floor(A) - decided with ONE
// magic-number addition rounded down: floor(256 A) lands in the mantissa, its high byte is the result and a low byte of 0 or 255
// marks the pixel as undecided.  Those (~0.8 % of natural pixels; every pixel of an exactly flat window, whose C lies within 1e-4 of
// an integer) are recomputed with the exact chain; a wave with too many of them recomputes its bands densely with the same exact
// code.  The output is bit-identical to the chain in every case.
#define BLUR_BAND 0.00390625f

// separable factors: gv[j] = exp(-j^2/200) and gh[k] = exp(-k^2/200) / 47.092777252197266 (the reference's f32 weight sum 0x423C5F01),
// rounded to f32 from double; sum |gv[j] gh[k] - w[j][k]| = 3.4e-8 for these.  Horizontal stage first (gh), then vertical (gv).
static __constant__ float c_sep_v[4] = {1.0f, 0.99501247919268232f, 0.98019867330675525f, 0.95599748183309996f};
static __constant__ float c_sep_h[4] = {(float)(1.0 / 47.092777252197266), (float)(0.99501247919268232 / 47.092777252197266),
                                 (float)(0.98019867330675525 / 47.092777252197266), (float)(0.95599748183309996 / 47.092777252197266)};

typedef unsigned blur_u4 __attribute__((ext_vector_type(4)));

// one workgroup of k_blur: image b, workgroup blk of the image's g.blur_blocks
__device__ __forceinline__ void blur_workgroup(const Geometry &g, const ImageSrc &src, const uint8_t *slab, uint8_t *blur_slab, const uint32_t *__restrict__ ctab, int b, int blk)
{
    __shared__ __align__(16) unsigned char s_mask[BLUR_THREADS * BLUR_RB_MAX];      // per lane: one byte per output row, bit k = pixel k undecided
    __shared__ unsigned short s_list[BLUR_THREADS / 64][BLUR_AMB_CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const unsigned wd = ctab_load(ctab, ctab_blur(g) + blk);      // level | workgroup of the level << 4
    const int lvl = (int)(wd & 15u), wb = (int)(wd >> 4);
    const LevelDesc &lv = g.lv[lvl];
    const int H = lv.H, W = lv.W, ncs = lv.blur_bx, RB = lv.blur_rb;
    asm volatile("" ::"s"(lv.img_off), "s"(lv.pitch), "s"(H), "s"(W), "s"(ncs), "s"(RB), "s"(lv.blur_by), "s"(lv.blur_recip));
    int pitch;
    const uint8_t *img = level_ptr_uniform(g, src, slab, b, lvl, lv.pitch, lv.img_off, pitch);
    uint8_t *const out_base = blur_slab + (size_t)b * g.slab_bytes + lv.img_off;
    const int out_pitch = lv.pitch;
    const int n_items = ncs * lv.blur_by;
    // bounds-checked buffer over the level plane: rows a short last band asks for beyond the image read as 0 and are never used
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img), 0, (unsigned)(H * pitch), 0x00020000);

    auto item_geometry = [&](int item, int &x0, int &ya) {
        const int band = ncs == 1 ? item : (int)__umulhi((unsigned)item, lv.blur_recip), strip = item - band * ncs;      // (2^32 / 1 does not fit the reciprocal)
        x0 = JSORB_BORDER - BLUR_X_LEAD + BLUR_SW * strip;
        ya = JSORB_BORDER + band * RB;
        return band;
    };
    const int item = wb * BLUR_THREADS + tid;
    const bool live = item < n_items;
    int x0, ya;
    const int my_band = item_geometry(live ? item : 0, x0, ya);
    // Walking direction of the lane: even bands top-down, odd bands BOTTOM-UP (round 6).  The six input rows around the boundary of two bands are read by
    // both; with every band walking down, the upper band reads them at the END of its walk and the lower one at the START - a whole wave lifetime
    // (~8 us = 36 MB of traffic at the kernel's rate, against 4 MB of L2 per XCD) apart, so they come from HBM twice: k_blur fetched 1.9 x its plane and
    // is bound by exactly that.  Walking towards each other, the two bands reach a shared boundary at the same time.  The vertical weights are
    // symmetric, the certificate's bound does not depend on the order of the 7-term chain, undecided pixels take the exact path either way.
#if BLUR_BOUSTRO
    const bool up = (my_band & 1) != 0;
#else
    const bool up = false;
#endif
    const int n_out = live ? min(RB, H - JSORB_BORDER - ya) : 0;                  // output rows of this lane
    const int n_valid = min(BLUR_SW, W - JSORB_BORDER - x0);                       // the strip's pixels up to the right end of the ROI (>= 1)
    const unsigned px_mask = (n_valid >= 8 ? 0xFFu : (1u << n_valid) - 1u) & (BLUR_X_LEAD && x0 < JSORB_BORDER ? 0xFFu << (JSORB_BORDER - x0) : 0xFFu);      // pixels inside the ROI

    // weights in vector registers (a scalar operand halves the issue rate of the 2-clock instructions)
    float gh[4], gv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        gh[k] = c_sep_h[k]; gv[k] = c_sep_v[k];
        asm volatile("" : "+v"(gh[k]), "+v"(gv[k]));
    }
    const float magic = 49152.0f;
    unsigned char *const my_mask = s_mask + tid * BLUR_RB_MAX;

    // ---- fast pass: stream down the band ----
    const int NR = RB + 6;                                                          // input rows ya - 3 .. ya + RB + 2 (wave-uniform count)
    const int in_step = up ? -pitch : pitch;                                        // the walk's row step in the input plane ...
    const unsigned out_step = (unsigned)(up ? -out_pitch : out_pitch);              // ... and in the output plane
    int off = (up ? ya + RB + 2 : ya - 3) * pitch + x0 - 4;                         // byte offset of the lane's 16-byte window: columns x0 - 4 .. x0 + 11 of the walk's first input row
    blur_u4 cur = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);             // (rows past the image - the short last band walking up - read as 0: bounds-checked buffer)
    // rows 1 .. BLUR_PREFETCH - 1 are requested up front as well (NR >= 7: they always exist); ahead[0] is the row after `cur`
    blur_u4 ahead[BLUR_PREFETCH > 1 ? BLUR_PREFETCH - 1 : 1];
#pragma unroll
    for (int q = 0; q + 1 < BLUR_PREFETCH; q++) { off += in_step; ahead[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); }
    float Hw[7][BLUR_SW];                                                           // horizontal sums of the last 7 input rows (slot = input row mod 7)
    unsigned n_amb_lane = 0;
    unsigned dst_row_off = (unsigned)((up ? ya + RB - 1 : ya) * out_pitch + x0);
    int o_real = up ? RB - 1 : 0;                                                   // band row of the walk's next output row
    const int o_step = up ? -1 : 1;
#pragma unroll 1
    for (int jb = 0; jb < NR; jb += 7) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int j = jb + u;
            if (j >= NR) break;                                                     // wave-uniform
            // BLUR_PREFETCH rows ahead of the one being evaluated are in flight: a row's arithmetic is ~170 ns, a load under a busy memory system takes
            // longer (round 6: one row ahead 0.189 ms per step, two rows 0.178 ms, C2 +1.5 % pairs/s)
            blur_u4 nxt = cur, far = cur;
            if (j + BLUR_PREFETCH < NR) { off += in_step; far = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); }
            if (BLUR_PREFETCH > 1) {
                nxt = ahead[0];
#pragma unroll
                for (int q = 0; q + 2 < BLUR_PREFETCH; q++) ahead[q] = ahead[q + 1];
                ahead[BLUR_PREFETCH - 2 >= 0 ? BLUR_PREFETCH - 2 : 0] = far;
            } else nxt = far;
            // 14 conversions: window bytes 1 .. 14 = columns x0 - 3 .. x0 + 10
            float f[14];
#pragma unroll
            for (int k = 0; k < 14; k++) {
                const unsigned wv = (k + 1) < 4 ? cur.x : (k + 1) < 8 ? cur.y : (k + 1) < 12 ? cur.z : cur.w;
                f[k] = (float)((wv >> (8 * ((k + 1) & 3))) & 0xFFu);
            }
            // horizontal stage: 8 sums of 7 taps, one chain per pixel (left to right)
#pragma unroll
            for (int k = 0; k < BLUR_SW; k++) {
                float t = gh[3] * f[k];
                t = __builtin_fmaf(gh[2], f[k + 1], t);
                t = __builtin_fmaf(gh[1], f[k + 2], t);
                t = __builtin_fmaf(gh[0], f[k + 3], t);
                t = __builtin_fmaf(gh[1], f[k + 4], t);
                t = __builtin_fmaf(gh[2], f[k + 5], t);
                Hw[u][k] = __builtin_fmaf(gh[3], f[k + 6], t);
            }
            cur = nxt;
            if (j < 6) continue;                                                    // wave-uniform: the window is not full yet
            // vertical stage of output row o = j - 6: window rows oldest .. newest = slots u + 1 .. u + 7 (mod 7)
            const int o = j - 6;
            float a[BLUR_SW];
#pragma unroll
            for (int k = 0; k < BLUR_SW; k++) {
                float t = gv[3] * Hw[(u + 1) % 7][k];
                t = __builtin_fmaf(gv[2], Hw[(u + 2) % 7][k], t);
                t = __builtin_fmaf(gv[1], Hw[(u + 3) % 7][k], t);
                t = __builtin_fmaf(gv[0], Hw[(u + 4) % 7][k], t);
                t = __builtin_fmaf(gv[1], Hw[(u + 5) % 7][k], t);
                t = __builtin_fmaf(gv[2], Hw[(u + 6) % 7][k], t);
                a[k] = __builtin_fmaf(gv[3], Hw[u][k], t);
            }
            // q = floor(256 A) in the low 16 mantissa bits of A + 49152 (ulp 2^-8) ROUNDED DOWN: the eight additions run with the wave's f32
            // rounding mode switched to -inf (everything else in this kernel is round-to-nearest-even).  Bits 8-15 are floor(A); a
            // fraction byte of 0 or 255 puts A within 2^-8 = 3.9e-3 of an integer, four times the bound on |A - C|: undecided.
            float r[BLUR_SW];
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 1\n\t"
                         "v_add_f32 %0, %8, %16\n\tv_add_f32 %1, %9, %16\n\tv_add_f32 %2, %10, %16\n\tv_add_f32 %3, %11, %16\n\t"
                         "v_add_f32 %4, %12, %16\n\tv_add_f32 %5, %13, %16\n\tv_add_f32 %6, %14, %16\n\tv_add_f32 %7, %15, %16\n\t"
                         "s_nop 1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                         : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(magic));
            unsigned ow[2], ambm = 0;
#pragma unroll
            for (int wi = 0; wi < 2; wi++) {
                const unsigned a01 = __builtin_amdgcn_perm(__float_as_uint(r[4 * wi + 1]), __float_as_uint(r[4 * wi]), 0x04000501u);
                const unsigned a23 = __builtin_amdgcn_perm(__float_as_uint(r[4 * wi + 3]), __float_as_uint(r[4 * wi + 2]), 0x04000501u);
                ow[wi] = __builtin_amdgcn_perm(a23, a01, 0x05040100u);                 // floor(A) of the four pixels
                const unsigned fr = __builtin_amdgcn_perm(a23, a01, 0x07060302u);       // floor(256 A) mod 256
                // y = (f ^ f << 1) & 0xFE is zero exactly for fraction bytes 0 and 255; ~y & (y - 0x01010101) has bit 7 of every zero byte of
                // y set (and possibly that of a byte of value 1 above one, which only sends a decided pixel through the exact code as well)
                const unsigned y = (fr ^ (fr << 1)) & 0xFEFEFEFEu;
                const unsigned z = (~y & (y - 0x01010101u)) & 0x80808080u;
                // bit 8t + 7 -> bit 4 wi + t: bytes of z are 0 or 128, one v_dot4_u32_u8 with the byte weights 1, 2, 4, 8 (16 .. 128 for the second dword) adds up
                // 128 x the nibble.  (Round 6: the 32-bit multiply this replaces - (z >> 7) * 0x01020408 >> 24 - is a quarter-rate instruction, two of them per
                // row of 8 pixels were a tenth of the kernel's issue time.)
                ambm = __builtin_amdgcn_udot4(z, wi ? 0x80402010u : 0x08040201u, ambm, false);
            }
            ambm >>= 7;
            const unsigned dst_off = dst_row_off;                                    // byte offset of this output row's 8 pixels from out_base (a running sum: no 64-bit multiply per row)
            dst_row_off += out_step;
            const int orow = o_real;
            o_real += o_step;
            if (orow < n_out) {
                uint8_t *dst = out_base + dst_off;
                if (BLUR_X_LEAD) {
                    // whole 8-byte units, zeros outside the ROI (the row's pitch has room for the last strip: pitch >= W rounded up to 64)
                    const unsigned long long keep = ((px_mask & 1u) ? 0xFFull : 0) | ((px_mask & 2u) ? 0xFF00ull : 0) | ((px_mask & 4u) ? 0xFF0000ull : 0) | ((px_mask & 8u) ? 0xFF000000ull : 0) |
                                                    ((px_mask & 16u) ? 0xFF00000000ull : 0) | ((px_mask & 32u) ? 0xFF0000000000ull : 0) | ((px_mask & 64u) ? 0xFF000000000000ull : 0) | ((px_mask & 128u) ? 0xFF00000000000000ull : 0);
                    *reinterpret_cast<unsigned long long *>(dst) = (((unsigned long long)ow[1] << 32) | ow[0]) & keep;
                } else if (n_valid >= BLUR_SW) { reinterpret_cast<unsigned *>(dst)[0] = ow[0]; reinterpret_cast<unsigned *>(dst)[1] = ow[1]; }
                else {
#pragma unroll
                    for (int k = 0; k < BLUR_SW; k++)
                        if (k < n_valid) dst[k] = (uint8_t)((ow[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                }
                ambm &= px_mask;
                my_mask[orow] = (unsigned char)ambm;
                n_amb_lane += __popc(ambm);
            }
        }
    }

    // ---- undecided pixels of the wave: list them (wave prefix sum of the per-lane counts, no atomics), recompute with the exact chain ----
    const int incl = wave_inclusive_scan_i32((int)n_amb_lane);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    // exact value of pixel (x, y): acc = fma(w[r][c], I, acc) in raster order, taps from the level plane (16 bytes from the dword below x - 3 per row)
    auto exact_pixel = [&](int x, int y) {
        const int sh = (x - 3) & 3;
        const int o16 = (y - 3) * pitch + ((x - 3) & ~3);
        blur_u4 v[7];
#pragma unroll
        for (int rr = 0; rr < 7; rr++) v[rr] = __builtin_amdgcn_raw_buffer_load_b128(rs, o16 + rr * pitch, 0, 0);      // all seven rows in flight together
        float acc = 0.0f;
#pragma unroll
        for (int rr = 0; rr < 7; rr++) {
            const unsigned w0 = __builtin_amdgcn_alignbyte(v[rr].y, v[rr].x, (unsigned)sh), w1 = __builtin_amdgcn_alignbyte(v[rr].z, v[rr].y, (unsigned)sh);
            const float g0 = c_gauss[rr][0], g1 = c_gauss[rr][1], g2 = c_gauss[rr][2], g3 = c_gauss[rr][3];
            acc = __builtin_fmaf(g3, (float)(w0 & 0xFFu), acc);
            acc = __builtin_fmaf(g2, (float)((w0 >> 8) & 0xFFu), acc);
            acc = __builtin_fmaf(g1, (float)((w0 >> 16) & 0xFFu), acc);
            acc = __builtin_fmaf(g0, (float)(w0 >> 24), acc);
            acc = __builtin_fmaf(g1, (float)(w1 & 0xFFu), acc);
            acc = __builtin_fmaf(g2, (float)((w1 >> 8) & 0xFFu), acc);
            acc = __builtin_fmaf(g3, (float)((w1 >> 16) & 0xFFu), acc);
        }
        out_base[(size_t)y * out_pitch + x] = (uint8_t)((unsigned)acc & 0xFFu);      // cvt.rzi.u32.f32 + st.u8
    };
    if (total <= BLUR_AMB_CAP) {
        unsigned short *my_list = s_list[wave];
        int pos = incl - (int)n_amb_lane;
        if (n_amb_lane) {
            // the lane's mask bytes as dwords (rows the lane never wrote - beyond n_out - are skipped): bit 8 (o & 3) + k of dword o >> 2 = pixel k of row o
            const blur_u4 *mq = reinterpret_cast<const blur_u4 *>(my_mask);
#pragma unroll
            for (int q = 0; q < BLUR_RB_MAX / 16; q++) {
                const blur_u4 mv = mq[q];
                const unsigned md[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
                for (int d4 = 0; d4 < 4; d4++) {
                    const int o0 = 16 * q + 4 * d4;
                    unsigned m = md[d4];
                    if (o0 + 4 > n_out) m &= o0 >= n_out ? 0u : (0xFFFFFFFFu >> (8 * (o0 + 4 - n_out)));
                    while (m) {
                        const int bit = __builtin_ctz(m);
                        m &= m - 1;
                        my_list[pos++] = (unsigned short)((lane << 8) | (o0 << 3) | bit);      // (o0 + bit / 8) << 3 | bit % 8
                    }
                }
            }
        }
        // (LDS operations of one wave execute in order: the list is complete for every lane of the wave here)
        const int item0 = wb * BLUR_THREADS + 64 * wave;
        for (int i = lane; i < total; i += 64) {
            const int e = my_list[i];
            int ex0, eya;
            item_geometry(item0 + (e >> 8), ex0, eya);
            exact_pixel(ex0 + (e & 7), eya + ((e >> 3) & 31));
        }
        return;
    }
    // ---- dense exact path (bands of mostly flat windows): every pixel of the lane's strip through the reference's chain ----
    for (int o = 0; o < n_out; o++)
        for (int k = 0; k < n_valid; k++)
            if ((px_mask >> k) & 1u) exact_pixel(x0 + k, ya + o);
}

} // namespace jsorb
