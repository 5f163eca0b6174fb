// k_stereo.hip - stereo matcher on device: row-epipolar candidate search + Hamming brute force + 11-shift 11x11 L1
// refinement + sub-pixel parabola + depth, one wave64 per up to 32 left keypoints (three phases, see k_stereo below), all pairs of a
// batch in ONE launch; then one workgroup per pair for the 2.1 x median outlier cut.
//
// Semantics restated (bit-exact): ORB_GPU::ORB_compute_stereo_match, src/cuda/orb_stereo_match.cu:105-580
//   :119-140  row table: right keypoint iR covers rows [floor(y-r), ceil(y+r)], r = 2*scale[octave]
//   :150-184  candidates of left keypoint: same row, |octave difference| <= 1, uR in [uL - maxD, uL]
//   :28-53    K12 Hamming (SWAR popcount == popcount); :241-256 arg-min, strict <, initial best = TH_HIGH
//   :282-325  keep if best < (TH_HIGH+TH_LOW)/2 ; coordinates scaled to the left octave with round(); bounds check
//   :64-102   K13 + cublasSgemv(:463): L1(s) = sum_{11x11} |(L - Lc) - (R_s - Rc_s)|, s in [-5,5] (exact integers)
//   :491-560  arg-min over the 11 shifts (reject the ends), parabola, bestuR, disparity test, depth = mbf / disparity
//   :563-578  sort by L1 distance, median, thDist = 1.5f*1.4f*median, remove everything >= thDist
// The reference crosses host<->device >= 12 times per frame here, with cudaMalloc/cudaFree and cublasCreate/Destroy
// inside the frame loop and M*1331 floats written to HBM only to be summed; this version never leaves the device.
// Candidates come from the scan-line buckets k_compact sorts the right keypoints into (one bucket per level and level-0 row: the
// reference's vRowIndices, :119-140, with every keypoint listed once); the run of buckets that covers the left keypoint's row is
// computed exactly per level, the reference's disparity-window test follows per candidate.  The tile-row start tables of k_compact
// (keypoints of one tile row are contiguous and ordered) remain as the fallback for geometries whose buckets do not fit k_compact's
// LDS (JSORB_STEREO_EPI=0 forces it).  Candidate order (ascending iR in the reference) only matters for ties, which the
// (distance << 20 | iR) min-key reproduces.
#include <algorithm>
#include <cstdlib>

#include "jsorb_launch.h"

namespace jsorb {

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

#define SKPW 4              // left keypoints per wave
#define SGL (64 / SKPW)     // lanes per left keypoint

__device__ __forceinline__ void wave_lds_sync_st()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Round 4 (second half): ONE wave64 handles up to 32 left keypoints in three phases.
//  A   candidate search, as before: 16 lanes per left keypoint, four keypoints per pass, `npass` passes (8 for batches; 1 for a single pair,
//      where the latency of a wave counts, not the number of instructions); the arg-min key and the candidate count of every keypoint go to LDS.
//  A2  lane = keypoint (32 lanes): everything the reference does once per left keypoint between K12 and K13 - match threshold, scaled
//      coordinates, the window-fits test - is issued ONCE per 32 keypoints instead of once per pass of four; the keypoints whose window search
//      runs are ranked by a ballot, their window addresses and the centre pixels of both windows (Lc, Rc_s) are staged in LDS.
//  B   the L1 sums as a FLAT list of (keypoint, window row) tasks, 64 per pass: a lane owns one row of one keypoint's window and evaluates
//      all 11 shifts as before; the 11 per-row sums are added into the keypoint's LDS counters two per ds_add_u32 (a window's sum is < 2^16).
//      Round 3 ran this phase with 11 of every 16 lanes, for unmatched keypoints too: 8 passes per 32 keypoints where 11 * n_refined / 64
//      (4.4 at the EuRoC shape, 4.9 KAIST-shaped) are needed.
//  C   lane = keypoint again: arg-min of the 11 sums, parabola, disparity test, depth, and one coalesced store per output array.
#define ST_MAX_PASS 8
#define ST_MAX_KP (SKPW * ST_MAX_PASS)
#ifndef ST_MIN_WAVES
#define ST_MIN_WAVES 5         // waves per SIMD the register allocation must allow (88 VGPRs as compiled: 5)
#endif
__global__ __launch_bounds__(64, ST_MIN_WAVES) void k_stereo(Geometry g, ImageSrc srcL, const uint8_t *slabL, ImageSrc srcR, const uint8_t *slabR,
                                               const int32_t *__restrict__ outL, const int *__restrict__ countsL, const uint8_t *__restrict__ descL,
                                               const int32_t *__restrict__ outR, const int *__restrict__ countsR, const uint8_t *__restrict__ descR,
                                               const int *__restrict__ row_tabR,
                                               float *__restrict__ u_right, float *__restrict__ depth, int *__restrict__ best_l1,
                                               unsigned *__restrict__ aux, StereoArgs sa, int n_pairs, int *__restrict__ diag, int npass)
{
    __shared__ int s_lvi[JSORB_MAX_LEVELS][12];      // th, nth, row_tab_off, W, pitch, img_off, 1/th magic, tw, ntw, tile_off, 1/tw magic (per level, lane-indexable)
    __shared__ float s_lvf[JSORB_MAX_LEVELS][2];     // scale, inv_scale
    __shared__ unsigned s_best[ST_MAX_KP];           // phase A -> A2: arg-min key of K12 per keypoint slot
    __shared__ int s_ncand[ST_MAX_KP];
    __shared__ __align__(16) unsigned s_par[ST_MAX_KP][8];      // per REFINED keypoint (rank): window row 0 of the left / right image (64-bit addresses), the two pitches, the byte shifts
    __shared__ __align__(16) unsigned s_ctr[ST_MAX_KP][4];      // centre row: W[1] of the left window, X[1..3] of the right band
    __shared__ __align__(16) unsigned s_acc[ST_MAX_KP][8];      // the 11 L1 sums, two per dword
    const int lane = threadIdx.x;
    const int grp = lane / SGL, sl = lane % SGL;
    const int kpw = SKPW * npass;                    // left keypoints of this wave
    int b, blk;
    if (!xcd_map((g.T + kpw - 1) / kpw, n_pairs, b, blk)) return;
    const int Nl = uniform_i32(countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS]);
    const int Nr = uniform_i32(countsR[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS]);
    if (blk * kpw >= Nl) return;                      // whole wave idle
    if (lane < g.L) {
        const LevelDesc &lv = g.lv[lane];
        s_lvi[lane][3] = lv.W; s_lvi[lane][4] = lv.pitch; s_lvi[lane][5] = (int)lv.img_off;
        if (!g.epi_rows) {                           // the tile-based candidate windows only
            s_lvi[lane][0] = lv.th; s_lvi[lane][1] = lv.nth; s_lvi[lane][2] = lv.row_tab_off;
            s_lvi[lane][6] = (int)(0xFFFFFFFFu / (unsigned)lv.th + 1u);      // x / th == umulhi(x, magic) for x < 2^16, th > 1
            s_lvi[lane][7] = lv.tw; s_lvi[lane][8] = lv.ntw; s_lvi[lane][9] = lv.tile_off;
            s_lvi[lane][10] = (int)(0xFFFFFFFFu / (unsigned)lv.tw + 1u);
        }
        s_lvf[lane][0] = lv.scale; s_lvf[lane][1] = lv.inv_scale;
    }
    if (lane < ST_MAX_KP) { s_best[lane] = 0xFFFFFFFFu; s_ncand[lane] = 0; }
    reinterpret_cast<uint4 *>(s_acc)[lane] = make_uint4(0, 0, 0, 0);          // 32 x 8 dwords = 64 x 16 bytes
    static_assert(ST_MAX_KP * 8 * 4 == 64 * 16, "one 16-byte store per lane clears the L1 counters");
    wave_lds_sync_st();
    const int32_t *oL = outL + (size_t)b * 6 * g.T;
    const int32_t *oR = outR + (size_t)b * 6 * g.T;
    const size_t tb = (size_t)b * g.T;

    // ---- phase A: candidate search, four left keypoints per pass ----
    // The left keypoint of the NEXT pass (coordinates, level, descriptor) is requested before the current pass starts its own chain of
    // dependent loads (bucket table -> entries -> right descriptors): one round trip less per pass.
    struct LeftKp { int x, y, level; uint4 a0, a1; };
    auto load_left = [&](int pass) {
        const int i = min(blk * kpw + pass * SKPW + grp, Nl - 1);      // idle groups shadow the last keypoint and report nothing
        const uint4 *dl = reinterpret_cast<const uint4 *>(descL + (tb + i) * 32);
        return LeftKp{oL[i], oL[Nl + i], oL[4 * (size_t)Nl + i], dl[0], dl[1]};
    };
#ifndef ST_PREFETCH
#define ST_PREFETCH 1
#endif
#if ST_PREFETCH
    LeftKp nxt = load_left(0);
#endif
    for (int pass = 0; pass < npass; pass++) {
        if (blk * kpw + pass * SKPW >= Nl) break;     // (wave-uniform)
        const int i_raw = blk * kpw + pass * SKPW + grp;
        const bool live = i_raw < Nl;
#if ST_PREFETCH
        const LeftKp cur = nxt;
        nxt = load_left(pass + 1);
#else
        const LeftKp cur = load_left(pass);
#endif
        const int levelL = cur.level;
        const float uL = (float)cur.x, vL = (float)cur.y;
        const float minU = uL - sa.maxD, maxU = uL - 0.0f;
        unsigned best_key = 0xFFFFFFFFu;
        int n_cand = 0;
        {
            const uint4 a0 = cur.a0, a1 = cur.a1;
            const int vLi = (int)vL;
            const int *rt = row_tabR + (size_t)b * g.row_tab_stride;
            if (g.epi_rows) {
                // Scan-line buckets (k_compact): a right keypoint of level lr in row y (its level-0 row, an integer) covers row vL iff
                // floor(y - r) <= vL <= ceil(y + r), r = 2 * scale[lr], both evaluated in f32 as the reference does.  Both bounds are
                // non-decreasing in y, so the rows that pass form ONE run [first, last] of that level's buckets: it is found exactly here,
                // once per level (lane t < 3 of the group takes level levelL - 1 + t and walks in from a bound that is two rows too wide on
                // each side), and the candidates need no row test of their own.  The three runs are then walked as one flat index space:
                // 16 entries of 8 bytes per step, coalesced.
                const int EH = g.epi_rows, EN = g.L * EH;
                const int *et = rt + g.epi_off;
                const int2 *ee = reinterpret_cast<const int2 *>(et + ((EN + 2) & ~1));
                int seg_start = 0, seg_len = 0;
                {
                    const int lr = levelL - 1 + sl;
                    if (sl < 3 && lr >= 0 && lr < g.L && !(maxU < 0)) {
                        const float r = 2.0f * s_lvf[lr][0], vLf = (float)vLi;
                        const float lo_f = __builtin_floorf(vLf - 2.0f - r), hi_f = __builtin_ceilf(vLf + 2.0f + r);
                        int below = 0, above = 0;                 // rows at the low / high end of [lo, hi] that do not cover vL
#pragma unroll
                        for (int k = 0; k < 5; k++) {
                            below += __builtin_ceilf((lo_f + (float)k) + r) < vLf ? 1 : 0;
                            above += __builtin_floorf((hi_f - (float)k) - r) > vLf ? 1 : 0;
                        }
                        const int ylo = max((int)lo_f + below, 0), yhi = min((int)hi_f - above, EH - 1);
                        if (ylo <= yhi) {
                            seg_start = et[lr * EH + ylo];
                            seg_len = et[lr * EH + yhi + 1] - seg_start;
                        }
                    }
                }
                const int gl0 = lane & ~(SGL - 1);
                const int st0 = __shfl(seg_start, gl0, 64), st1 = __shfl(seg_start, gl0 + 1, 64), st2 = __shfl(seg_start, gl0 + 2, 64);
                const int c1 = __shfl(seg_len, gl0, 64), c2 = c1 + __shfl(seg_len, gl0 + 1, 64), total = c2 + __shfl(seg_len, gl0 + 2, 64);
                // The kernel waits on its dependent loads (entry -> descriptor), not on its ALUs: three steps of 16 entries are requested
                // together, the disparity-window test runs on the entries alone, and the three descriptor loads go out together as well
                // (lanes whose entry fails the test read descriptor 0: one cache line for all of them) - two round trips per 48
                // candidates instead of six.
#ifndef ST_CAND_UNROLL
#define ST_CAND_UNROLL 3
#endif
                constexpr int CU = ST_CAND_UNROLL;
                for (int k = sl; k < total; k += CU * SGL) {
                    int2 e[CU];
                    bool ok[CU];
#pragma unroll
                    for (int u = 0; u < CU; u++) {
                        const int kk = k + u * SGL;
                        const int kc = kk < total ? kk : k;
                        e[u] = ee[(kc >= c2 ? st2 - c2 : (kc >= c1 ? st1 - c1 : st0)) + kc];
                    }
                    uint4 b0[CU], b1[CU];
#pragma unroll
                    for (int u = 0; u < CU; u++) {
                        const float uR = (float)(e[u].y & 0xFFFF);
                        ok[u] = k + u * SGL < total && uR >= minU && uR <= maxU;
                        const uint4 *dr = reinterpret_cast<const uint4 *>(descR + (tb + (ok[u] ? e[u].x : 0)) * 32);
                        b0[u] = dr[0]; b1[u] = dr[1];
                    }
#pragma unroll
                    for (int u = 0; u < CU; u++) {
                        if (!ok[u]) continue;
                        n_cand++;
                        const int d = hamming256(a0, a1, b0[u], b1[u]);
                        if (d < sa.th_high) {
                            const unsigned key = ((unsigned)d << 20) | (unsigned)e[u].x;
                            best_key = key < best_key ? key : best_key;
                        }
                    }
                }
            } else {
            int j0[3], len[3];
            int nrw[3], tl0[3], tst[3], ncl[3];      // column-pruned form: tile rows of the band, tile index of (first row, first column), tiles per row, columns of the window
            float rr[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int lr = levelL - 1 + t;
                j0[t] = 0; len[t] = 0; rr[t] = 0.f;
                nrw[t] = 0; tl0[t] = 0; tst[t] = 0; ncl[t] = 0;
                if (lr >= 0 && lr < g.L && !(maxU < 0)) {
                    const float scl = s_lvf[lr][0], iscl = s_lvf[lr][1];
                    const int th = s_lvi[lr][0], nth = s_lvi[lr][1], rto = s_lvi[lr][2];
                    const unsigned th_magic = (unsigned)s_lvi[lr][6];
                    const float r = 2.0f * scl;
                    rr[t] = r;
                    // conservative tile-row window of level lr (exact tests follow per candidate): the product with 1/scale instead of
                    // the quotient moves the bounds by ~1e-4 px, the window carries a margin of a whole pixel on each side
                    const int lo = (int)__builtin_floorf((vL - r - 1.0f) * iscl) - 1;
                    const int hi = (int)__builtin_ceilf((vL + r + 2.0f) * iscl) + 1;
                    const int t_lo = lo < 0 ? 0 : (th > 1 ? (int)__umulhi((unsigned)lo, th_magic) : lo);
                    int t_hi = hi < 0 ? -1 : (th > 1 ? (int)__umulhi((unsigned)hi, th_magic) : hi);
                    if (t_hi > nth - 1) t_hi = nth - 1;
                    if (t_lo <= t_hi) {
                        if (!g.stereo_colprune) {
                            j0[t] = rt[rto + t_lo];
                            len[t] = rt[rto + t_hi + 1] - j0[t];
                        } else {
                            const int tw = s_lvi[lr][7], ntw = s_lvi[lr][8];
                            const unsigned tw_magic = (unsigned)s_lvi[lr][10];
                            const int xlo = (int)__builtin_floorf((minU - 1.0f) * iscl) - 1, xhi = (int)__builtin_ceilf((maxU + 1.0f) * iscl) + 1;
                            const int c_lo = xlo < 0 ? 0 : (tw > 1 ? (int)__umulhi((unsigned)xlo, tw_magic) : xlo);
                            int c_hi = xhi < 0 ? -1 : (tw > 1 ? (int)__umulhi((unsigned)xhi, tw_magic) : xhi);
                            if (c_hi > ntw - 1) c_hi = ntw - 1;
                            if (c_lo <= c_hi) {
                                nrw[t] = t_hi - t_lo + 1;
                                tl0[t] = s_lvi[lr][9] + t_lo * ntw + c_lo;
                                tst[t] = ntw;
                                ncl[t] = c_hi - c_lo + 1;
                            }
                        }
                    }
                }
            }
            // one right keypoint against this left keypoint: the reference's exact row / column tests, then the Hamming distance
            auto candidate_at = [&](int j, float kpY, float uR, float r) {
                const uint4 *dr = reinterpret_cast<const uint4 *>(descR + (tb + j) * 32);
                const uint4 b0 = dr[0], b1 = dr[1];
                const int maxr = (int)__builtin_ceilf(kpY + r), minr = (int)__builtin_floorf(kpY - r);
                if (vLi < minr || vLi > maxr) return;
                if (!(uR >= minU && uR <= maxU)) return;
                n_cand++;
                const int d = hamming256(a0, a1, b0, b1);
                if (d < sa.th_high) {
                    const unsigned key = ((unsigned)d << 20) | (unsigned)j;
                    best_key = key < best_key ? key : best_key;
                }
            };
            auto candidate = [&](int j, float r) { candidate_at(j, (float)oR[Nr + j], (float)oR[j], r); };
            if (!g.stereo_colprune) {
                const int c1 = len[0], c2 = len[0] + len[1], total = c2 + len[2];
                for (int k = sl; k < total; k += SGL) {
                    const int t = k >= c2 ? 2 : (k >= c1 ? 1 : 0);
                    const int j = (t == 2 ? j0[2] - c2 : (t == 1 ? j0[1] - c1 : j0[0])) + k;
                    candidate(j, t == 2 ? rr[2] : (t == 1 ? rr[1] : rr[0]));
                }
            } else {
                // Column pruning: the keypoints of a tile row are ordered by tile column (one per tile at most), so those inside the disparity
                // window [uL - maxD, uL] sit in a contiguous run of tiles.  Every (level, tile row) of the band becomes a SEGMENT
                // [tile_pos[row, c_lo], tile_pos[row, c_hi + 1]) of the right image's keypoint list (per-tile start table of k_compact); lane s of
                // the group fetches segment s (one round trip for all of them), then the group walks the segments.  At 64 tiles per row and a
                // window of 23 tiles this scans a third of what the whole-row form scans.  The window is conservative by a pixel on each side
                // (a keypoint's level-0 x is int(x_level * scale)); the exact tests above decide.
                const int *tp = rt + g.row_tab_len;
                const int c1 = nrw[0], c2 = nrw[0] + nrw[1], nseg = c2 + nrw[2];
                int nseg_max = nseg;                         // largest segment count of the wave's four keypoints (wave-uniform)
                nseg_max = max(max(__builtin_amdgcn_readlane(nseg, 0), __builtin_amdgcn_readlane(nseg, 16)),
                               max(__builtin_amdgcn_readlane(nseg, 32), __builtin_amdgcn_readlane(nseg, 48)));
                for (int s0 = 0; s0 < nseg_max; s0 += SGL) {
                    const int sg = s0 + sl;
                    int seg_start = 0, seg_len = 0;
                    float seg_r = 0.f;
                    if (sg < nseg) {
                        const int t = sg >= c2 ? 2 : (sg >= c1 ? 1 : 0);
                        const int row = sg - (t == 2 ? c2 : (t == 1 ? c1 : 0));
                        const int tile = (t == 2 ? tl0[2] : (t == 1 ? tl0[1] : tl0[0])) + row * (t == 2 ? tst[2] : (t == 1 ? tst[1] : tst[0]));
                        seg_start = tp[tile];
                        seg_len = tp[tile + (t == 2 ? ncl[2] : (t == 1 ? ncl[1] : ncl[0]))] - seg_start;
                        seg_r = t == 2 ? rr[2] : (t == 1 ? rr[1] : rr[0]);
                    }
                    const int n_here = min(SGL, nseg_max - s0);
                    for (int q = 0; q < n_here; q++) {
                        const int src_lane = (lane & ~(SGL - 1)) + q;
                        const int st = __shfl(seg_start, src_lane, 64), ln = __shfl(seg_len, src_lane, 64);
                        const float r = __shfl(seg_r, src_lane, 64);
                        for (int k = sl; k < ln; k += SGL) candidate(st + k, r);
                    }
                }
            }
            }
        }
        static_assert(SGL == 16, "the lane group of a keypoint is one DPP row");
        best_key = row16_min_u32(best_key);           // reduce inside the keypoint's lane group
        n_cand = row16_sum_i32(n_cand);
        if (live && sl == 0) { s_best[pass * SKPW + grp] = best_key; s_ncand[pass * SKPW + grp] = n_cand; }
    }
    wave_lds_sync_st();

    // ---- phase A2: lane = keypoint slot (lanes 32..63 mirror 0..31 and stay passive) ----
    const int slot = lane & (ST_MAX_KP - 1);
    const int i_raw = blk * kpw + slot;
    const bool live = lane < ST_MAX_KP && slot < kpw && i_raw < Nl;
    const int i = min(i_raw, Nl - 1);
    const unsigned best_key = s_best[slot];
    const int n_cand = s_ncand[slot];
    const int xL0 = oL[i], yL0 = oL[Nl + i], levelL = oL[4 * (size_t)Nl + i];
    const float uL = (float)xL0, vL = (float)yL0;
    float out_u = -1.0f, out_d = -1.0f;
    int out_l1 = -1, corr = 0;
    const bool matched = best_key != 0xFFFFFFFFu && (int)(best_key >> 20) < sa.th_orb;
    const int bestIdxR = matched ? (int)(best_key & 0xFFFFFu) : 0;
    const float scL = s_lvf[levelL][0], iscL = s_lvf[levelL][1];
    const int WL = s_lvi[levelL][3];
    const float uR0 = (float)oR[bestIdxR];
    const float scaleduR0 = roundf(uR0 * iscL);
    const float scaleduL0 = roundf(uL * iscL);
    const float scaledvL0 = roundf(vL * iscL);
    const float iniu = scaleduR0 - 5.0f - 5.0f, endu = scaleduR0 + 5.0f + 5.0f;
    const bool refine = live && matched && !(iniu < 0 || endu >= (float)WL);
    const unsigned long long ref_mask = __ballot(refine);
    const int n_ref = __popcll(ref_mask);
    const int rank = __popcll(ref_mask & ((1ull << lane) - 1ull));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    if (refine) {
        corr = 1;
        int pl, pr;
        const uint8_t *imL, *imR;
        if (levelL == 0) {
            pl = srcL.l0_pitch; imL = srcL.l0 + (size_t)b * srcL.l0_stride;
            pr = srcR.l0_pitch; imR = srcR.l0 + (size_t)b * srcR.l0_stride;
        } else {
            pl = pr = s_lvi[levelL][4];
            imL = slabL + (size_t)b * g.slab_bytes + (unsigned)s_lvi[levelL][5];
            imR = slabR + (size_t)b * g.slab_bytes + (unsigned)s_lvi[levelL][5];
        }
        // the 11x16 B (left) / 11x32 B (right) that cover the window rows are read straight from the image (dword aligned addresses,
        // never past row y+5 <= H-16) and byte-aligned with v_alignbyte.  No LDS staging of pixels: a misaligned LDS read costs 64 clk
        // per wave-instruction on gfx950 - the first version of this phase spent most of its time there.
        const int xl = (int)scaleduL0, xr = (int)scaleduR0, y = (int)scaledvL0;
        const int la = (xl - 5) & ~3, ra = (xr - 10) & ~3;      // dword aligned window starts
        const unsigned shl = (unsigned)(xl - 5 - la), shr = (unsigned)(xr - 10 - ra);
        const uint8_t *aL = imL + (size_t)(y - 5) * pl + la, *aR = imR + (size_t)(y - 5) * pr + ra;
        const u32x4 Lq = *reinterpret_cast<const u32x4 *>(aL + (size_t)5 * pl);
        const u32x4 Rq0 = *reinterpret_cast<const u32x4 *>(aR + (size_t)5 * pr);
        const u32x4 Rq1 = *reinterpret_cast<const u32x4 *>(aR + (size_t)5 * pr + 16);
        const unsigned long long uaL = (unsigned long long)(uintptr_t)aL, uaR = (unsigned long long)(uintptr_t)aR;
        reinterpret_cast<uint4 *>(s_par[rank])[0] = make_uint4((unsigned)uaL, (unsigned)(uaL >> 32), (unsigned)uaR, (unsigned)(uaR >> 32));
        reinterpret_cast<uint4 *>(s_par[rank])[1] = make_uint4((unsigned)pl, (unsigned)pr, shl, shr);
        reinterpret_cast<uint4 *>(s_ctr[rank])[0] = make_uint4(__builtin_amdgcn_alignbyte(Lq.z, Lq.y, shl), __builtin_amdgcn_alignbyte(Rq0.z, Rq0.y, shr),
                                                               __builtin_amdgcn_alignbyte(Rq0.w, Rq0.z, shr), __builtin_amdgcn_alignbyte(Rq1.x, Rq0.w, shr));
    }
    wave_lds_sync_st();

    // ---- phase B: L1(s) = sum over the 11x11 window of |(L - Lc) - (R_s - Rc_s)| for the 11 shifts s, as (keypoint, row) tasks ----
    // A lane owns one window row: all 11 shifts with packed 16-bit SADs, |L - (R + k_s)| with k_s = Lc - Rc_s, both sides biased by 256
    // so that they stay positive.
    // (Round 6 measured the rows of the NEXT 64 tasks requested before the current ones are evaluated: k_stereo 0.101 -> 0.104 ms per step - dropped.)
    const int n_tasks = 11 * n_ref;
    for (int t0 = 0; t0 < n_tasks; t0 += 64) {
        const int t = t0 + lane;
        const bool mine = t < n_tasks;
        const int tt = mine ? t : 0;
        const int ridx = (tt * 745) >> 13;            // tt / 11 for tt < 2700
        const int row = tt - 11 * ridx;
        const uint4 p0 = reinterpret_cast<const uint4 *>(s_par[ridx])[0], p1 = reinterpret_cast<const uint4 *>(s_par[ridx])[1];
        const uint4 ctr = reinterpret_cast<const uint4 *>(s_ctr[ridx])[0];
        const uint8_t *rowL = reinterpret_cast<const uint8_t *>((uintptr_t)(((unsigned long long)p0.y << 32) | p0.x)) + (size_t)row * (int)p1.x;
        const uint8_t *rowR = reinterpret_cast<const uint8_t *>((uintptr_t)(((unsigned long long)p0.w << 32) | p0.z)) + (size_t)row * (int)p1.y;
        const unsigned shl = p1.z, shr = p1.w;
        u32x4 Lq = (u32x4){0, 0, 0, 0}, Rq0 = Lq, Rq1 = Lq;
        if (mine) {
            Lq = *reinterpret_cast<const u32x4 *>(rowL);
            Rq0 = *reinterpret_cast<const u32x4 *>(rowR);
            Rq1 = *reinterpret_cast<const u32x4 *>(rowR + 16);
        }
        // window bytes 0..10 of the row in W[0..2]; search band bytes 0..20 in X[0..5]
        unsigned W[3], X[6];
        W[0] = __builtin_amdgcn_alignbyte(Lq.y, Lq.x, shl); W[1] = __builtin_amdgcn_alignbyte(Lq.z, Lq.y, shl);
        W[2] = __builtin_amdgcn_alignbyte(Lq.w, Lq.z, shl);
        X[0] = __builtin_amdgcn_alignbyte(Rq0.y, Rq0.x, shr); X[1] = __builtin_amdgcn_alignbyte(Rq0.z, Rq0.y, shr);
        X[2] = __builtin_amdgcn_alignbyte(Rq0.w, Rq0.z, shr); X[3] = __builtin_amdgcn_alignbyte(Rq1.x, Rq0.w, shr);
        X[4] = __builtin_amdgcn_alignbyte(Rq1.y, Rq1.x, shr); X[5] = __builtin_amdgcn_alignbyte(Rq1.z, Rq1.y, shr);
        // centre row (window row 5) of both images: Lc = byte 5 of its W, Rc_s = byte 5+s of its X - staged per keypoint in phase A2
        const unsigned cW1 = ctr.x, cX1 = ctr.y, cX2 = ctr.z, cX3 = ctr.w;
        const unsigned cX[4] = {0u, cX1, cX2, cX3};
        const int lc = (int)((cW1 >> 8) & 0xFFu);
        // 16-bit pairs: A[m] = (L[2m], L[2m+1]) + 256 ; E[m] = (R[2m], R[2m+1]) ; O[m] = (R[2m+1], R[2m+2])
        unsigned A[6], E[11], O[10];
#pragma unroll
        for (int m = 0; m < 6; m++)
            A[m] = __builtin_amdgcn_perm(0u, W[m >> 1], (m & 1) ? 0x0c030c02u : 0x0c010c00u) + 0x01000100u;
        A[5] &= 0x0000FFFFu;                                      // the window has 11 columns: the 12th half-pair is masked
#pragma unroll
        for (int m = 0; m < 11; m++) E[m] = __builtin_amdgcn_perm(0u, X[m >> 1], (m & 1) ? 0x0c030c02u : 0x0c010c00u);
#pragma unroll
        for (int m = 0; m < 10; m++)
            O[m] = (m & 1) ? __builtin_amdgcn_perm(X[(m + 1) >> 1], X[(m - 1) >> 1], 0x0c040c03u) : __builtin_amdgcn_perm(0u, X[m >> 1], 0x0c020c01u);
        unsigned part[12];
        part[11] = 0;
#pragma unroll
        for (int q = 0; q < 11; q++) {
            const int rc = (int)((cX[(5 + q) >> 2] >> (8 * ((5 + q) & 3))) & 0xFFu);
            const unsigned kv = (unsigned)(lc - rc + 256);         // in [1, 511]
            const unsigned kpk = (kv << 16) | kv;
            unsigned pq_ = 0;
#pragma unroll
            for (int m = 0; m < 6; m++) {
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const unsigned Rp = (q & 1) ? O[(q >> 1) + m] : E[(q >> 1) + m];
                unsigned Bp = __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, Rp) + __builtin_bit_cast(us2, kpk)));
                if (m == 5) Bp &= 0x0000FFFFu;
                pq_ = __builtin_amdgcn_sad_u16(A[m], Bp, pq_);
            }
            part[q] = pq_;
        }
        // a row's partial sum is < 11 * 510 and a window's sum <= 121 * 510 = 61710 < 2^16: two shifts share a dword and an LDS addition
        if (mine) {
#pragma unroll
            for (int q2 = 0; q2 < 6; q2++) atomicAdd(&s_acc[ridx][q2], part[2 * q2] | (part[2 * q2 + 1] << 16));
        }
    }
    wave_lds_sync_st();

    // ---- phase C: lane = keypoint slot again ----
    int acc[11];
    {
        const uint4 a0 = reinterpret_cast<const uint4 *>(s_acc[refine ? rank : 0])[0], a1 = reinterpret_cast<const uint4 *>(s_acc[refine ? rank : 0])[1];
        acc[0] = (int)(a0.x & 0xFFFFu); acc[1] = (int)(a0.x >> 16); acc[2] = (int)(a0.y & 0xFFFFu); acc[3] = (int)(a0.y >> 16);
        acc[4] = (int)(a0.z & 0xFFFFu); acc[5] = (int)(a0.z >> 16); acc[6] = (int)(a0.w & 0xFFFFu); acc[7] = (int)(a0.w >> 16);
        acc[8] = (int)(a1.x & 0xFFFFu); acc[9] = (int)(a1.x >> 16); acc[10] = (int)(a1.y & 0xFFFFu);
    }
    if (refine) {
        // first minimum of the 11 sums (strict < in ascending order, :491-505) = minimum of the keys (sum << 4 | shift); sums are < 2^16
        unsigned kmin = ((unsigned)acc[0] << 4);
#pragma unroll
        for (int s = 1; s < 11; s++) kmin = min(kmin, ((unsigned)acc[s] << 4) | (unsigned)s);
        const int bestDist = (int)(kmin >> 4), bestR = (int)(kmin & 15u);
        if (!(bestR == 0 || bestR == 10)) {
            float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
            for (int s = 1; s < 10; s++)
                if (s == bestR) { dist1 = (float)acc[s - 1]; dist2 = (float)acc[s]; dist3 = (float)acc[s + 1]; }
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (!(deltaR < -1 || deltaR > 1)) {
                float bestuR = scL * ((scaleduR0 + (float)bestR - 5.0f) + deltaR);
                float disparity = uL - bestuR;
                if (disparity >= 0.0f && disparity < sa.maxD) {
                    if (disparity <= 0) {
                        disparity = 0.01f;
                        bestuR = (float)((double)uL - 0.01);
                    }
                    out_d = sa.mbf / disparity;
                    out_u = bestuR;
                    out_l1 = bestDist;
                }
            }
        }
    }
    if (live) {
        u_right[tb + i] = out_u;
        depth[tb + i] = out_d;
        best_l1[tb + i] = out_l1;
        // per-keypoint statistics; k_median reduces them per pair (per-wave global atomics on one cache line per pair
        // serialised at the L2 atomic unit and cost more than the whole matcher)
        aux[tb + i] = (n_cand & 0x7FFFFFFF) | (corr ? 0x80000000u : 0u);
        if (diag) {
            // inspection (jsorb_set_stereo_diagnostics): the intermediate results the reference keeps per left keypoint - best right index and
            // Hamming distance of K12's arg-min (orb_stereo_match.cu:241-256: -1 / th_high when no candidate is closer than th_high) and the 11
            // L1 sums of K13 + gemv (:294-470; only meaningful where the window search ran: bit 31 of aux)
            int *dg = diag + (size_t)(tb + i) * JSORB_STEREO_DIAG_INTS;
            const int bd = best_key == 0xFFFFFFFFu ? 0x7FFFFFFF : (int)(best_key >> 20);
            dg[0] = bd < sa.th_high ? (int)(best_key & 0xFFFFFu) : -1;
            dg[1] = bd < sa.th_high ? bd : sa.th_high;
#pragma unroll
            for (int q = 0; q < 11; q++) dg[2 + q] = corr ? acc[q] : -1;
        }
    }
}


// 2.1 x median cut (orb_stereo_match.cu:563-578).  The median of the sorted (dist, idx) pairs is the (nv/2)-th smallest
// distance; L1 distances are < 2^16 (121*510), so a two-pass 256-bin radix select in LDS finds it exactly.
// generic form (any number of keypoints): three passes over the distances in memory, bin searches by thread 0
__device__ __forceinline__ int block256_exclusive_scan(int v, int *s_w, int &total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int incl = wave_inclusive_scan_i32(v);
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    const int w0 = s_w[0], w1 = s_w[1], w2 = s_w[2], w3 = s_w[3];
    total = w0 + w1 + w2 + w3;
    const int base = wave == 0 ? 0 : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2;
    __syncthreads();                                       // s_w may be reused by the next call
    return base + incl - v;
}

// The (nv/2)-th smallest L1 distance by two histogram passes: bin of the high byte, then of the low byte inside that bin.  An L1 distance
// is a sum of 121 terms |(L - L_centre) - (R - R_centre)| <= 510, i.e. < 2^16.  The distances of a pair cluster in a few bins, where LDS
// atomics on one address serialise, so every wave adds into its own copy of the histogram (4 x 256 counters).
#define MED_BIN_SHIFT 8
static_assert((2 * 5 + 1) * (2 * 5 + 1) * 510 < (256 << MED_BIN_SHIFT), "L1 distance range exceeds the first-level histogram");

struct MedianShared {
    int hist[4][256];
    int sel[4];
    int s_w[4];
    int s_cand, s_corr, s_removed;
};

// VALUES(fn): calls fn(d, r) for every distance slot of this thread (d < 0: no match); the register-resident and the re-reading form
// share everything else.
template <class ForEach>
__device__ __forceinline__ float median_threshold(MedianShared &sm, int tid, ForEach for_each, int &nv_out)
{
    const int wave = uniform_i32(tid >> 6);
    for_each([&](int d) { if (d >= 0) atomicAdd(&sm.hist[wave][d >> MED_BIN_SHIFT], 1); });
    __syncthreads();
    int nv;
    {
        const int h = sm.hist[0][tid] + sm.hist[1][tid] + sm.hist[2][tid] + sm.hist[3][tid];
        const int excl = block256_exclusive_scan(h, sm.s_w, nv);
        const int kth = nv / 2;
        if (h > 0 && excl <= kth && kth < excl + h) { sm.sel[0] = tid; sm.sel[1] = kth - excl; }      // exactly one bin (nv > 0)
    }
    nv_out = nv;
    if (nv == 0) return 3.0e38f;                       // Appendix C-6: nothing matched -> no cut (nv is workgroup-uniform)
    __syncthreads();                                   // everybody has read the first-level counters
#pragma unroll
    for (int w = 0; w < 4; w++) sm.hist[w][tid] = 0;
    __syncthreads();
    const int bin = sm.sel[0], kin = sm.sel[1];
    for_each([&](int d) { if (d >= 0 && (d >> MED_BIN_SHIFT) == bin) atomicAdd(&sm.hist[wave][d & ((1 << MED_BIN_SHIFT) - 1)], 1); });
    __syncthreads();
    {
        const int h = sm.hist[0][tid] + sm.hist[1][tid] + sm.hist[2][tid] + sm.hist[3][tid];
        int tot;
        const int excl = block256_exclusive_scan(h, sm.s_w, tot);
        if (h > 0 && excl <= kin && kin < excl + h) sm.sel[3] = (bin << MED_BIN_SHIFT) | tid;
    }
    __syncthreads();
    const float median = (float)sm.sel[3];
    return 1.5f * 1.4f * median;
}

// Pairs with more left keypoints than the register-resident form holds: the distances are re-read from global memory (L2 hits) in
// each of the three passes.
__device__ __noinline__ void median_big(MedianShared &sm, const Geometry &g, const int *__restrict__ countsL, float *__restrict__ u_right,
                                        float *__restrict__ depth, const int *__restrict__ best_l1,
                                        const unsigned *__restrict__ aux, int *__restrict__ stats, DeliverStereo dl)
{
    const int tid = threadIdx.x, b = blockIdx.x;
    const int Nl = countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    const size_t tb = (size_t)b * g.T;
    int *st = stats + b * 8;
    int cand = 0, corr = 0;
    for (int i = tid; i < Nl; i += 256) {
        const unsigned a = aux[tb + i];
        cand += (int)(a & 0x7FFFFFFFu);
        corr += (int)(a >> 31);
    }
    cand = wave_sum_i32(cand);
    corr = wave_sum_i32(corr);
    if ((tid & 63) == 0) { atomicAdd(&sm.s_cand, cand); atomicAdd(&sm.s_corr, corr); }
    int nv;
    const float thDist = median_threshold(sm, tid, [&](auto f) { for (int i = tid; i < Nl; i += 256) f(best_l1[tb + i]); }, nv);
    int removed = 0;
    for (int i = tid; i < Nl; i += 256) {
        const int d = best_l1[tb + i];
        float u = -1.0f, z = -1.0f;
        if (dl.u_host) { u = u_right[tb + i]; z = depth[tb + i]; }
        if (d >= 0 && !((float)d < thDist)) {
            u = -1.0f; z = -1.0f;
            u_right[tb + i] = u;
            depth[tb + i] = z;
            removed++;
        }
        if (dl.u_host) { dl.u_host[i] = u; dl.d_host[i] = z; }
    }
    if (removed) atomicAdd(&sm.s_removed, removed);
    __syncthreads();
    if (tid == 0) {
        const int v[4] = {sm.s_cand, sm.s_corr, nv, nv - sm.s_removed};
#pragma unroll
        for (int k = 0; k < 4; k++) { st[k] = v[k]; if (dl.stats_host) dl.stats_host[b * 8 + k] = v[k]; }
    }
}

// Median cut of one stereo pair per workgroup (orb_stereo_match.cu:560-580: sort, median = dist[size/2], thDist = 1.5*1.4*median, matches
// with dist >= thDist lose uRight / depth) + the per-pair statistics.  Pairs with at most 256 * MED_R left keypoints keep their L1
// distances in registers over the three passes.
// Two builds: 32 distances per thread (pairs of up to 8 192 left keypoints: every shipped configuration but the KAIST-shaped one) and 64
// (up to 16 384: the KAIST-shaped pairs hold 12.8 k, and the three-pass re-reading form took 71 us per launch there); the launch picks by the
// handle's keypoint capacity, the kernel still falls back to median_big() when a pair exceeds its build.
template <int MED_R>
__global__ __launch_bounds__(256) void k_median(Geometry g, const int *__restrict__ countsL, float *__restrict__ u_right,
                                                float *__restrict__ depth, const int *__restrict__ best_l1,
                                                const unsigned *__restrict__ aux, int *__restrict__ stats, DeliverStereo dl)
{
    __shared__ MedianShared sm;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int Nl = countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    const size_t tb = (size_t)b * g.T;
    int *st = stats + b * 8;
#pragma unroll
    for (int w = 0; w < 4; w++) sm.hist[w][tid] = 0;
    if (tid == 0) { sm.s_cand = 0; sm.s_corr = 0; sm.s_removed = 0; }
    __syncthreads();
    if (Nl > 256 * MED_R) {                                // workgroup-uniform: too many keypoints for the register-resident form
        median_big(sm, g, countsL, u_right, depth, best_l1, aux, stats, dl);
        return;
    }
    int d[MED_R];
    int cand = 0, corr = 0;
#pragma unroll
    for (int r = 0; r < MED_R; r++) {
        const int i = tid + 256 * r;
        d[r] = -1;
        if (i < Nl) {
            d[r] = best_l1[tb + i];
            const unsigned a = aux[tb + i];
            cand += (int)(a & 0x7FFFFFFFu);
            corr += (int)(a >> 31);
        }
    }
    // Single-pair call: the final uRight / depth also go to the pinned host mirror, from THIS kernel only (one writer per host address: two
    // kernels storing to the same pinned word - k_stereo the value, this one the cut - would rely on the order of posted PCIe writes of
    // different kernels).  The values are requested here, with the distances, so that their latency hides behind the histogram passes
    // (requested in the delivery loop they were a chain of dependent round trips: 14.6 us for one pair instead of 11).
    float uu[MED_R], zz[MED_R];
    if (dl.u_host) {
#pragma unroll
        for (int r = 0; r < MED_R; r++) {
            const int i = tid + 256 * r;
            uu[r] = -1.0f; zz[r] = -1.0f;
            if (i < Nl) { uu[r] = u_right[tb + i]; zz[r] = depth[tb + i]; }
        }
    }
    cand = wave_sum_i32(cand);
    corr = wave_sum_i32(corr);
    if ((tid & 63) == 0) { atomicAdd(&sm.s_cand, cand); atomicAdd(&sm.s_corr, corr); }
    int nv;
    const float thDist = median_threshold(sm, tid, [&](auto f) {
#pragma unroll
        for (int r = 0; r < MED_R; r++) f(d[r]);
    }, nv);
    int removed = 0;
#pragma unroll
    for (int r = 0; r < MED_R; r++) {
        const int i = tid + 256 * r;
        const bool cut = d[r] >= 0 && !((float)d[r] < thDist);
        if (cut) {
            u_right[tb + i] = -1.0f;
            depth[tb + i] = -1.0f;
            removed++;
        }
        if (dl.u_host && i < Nl) { dl.u_host[i] = cut ? -1.0f : uu[r]; dl.d_host[i] = cut ? -1.0f : zz[r]; }      // ONE writer per pinned host word (a two-writer delivery was measured and dropped: profiles/r05_two_writer.txt)
    }
    if (removed) atomicAdd(&sm.s_removed, removed);
    __syncthreads();
    if (tid == 0) {
        const int v[4] = {sm.s_cand, sm.s_corr, nv, nv - sm.s_removed};
#pragma unroll
        for (int k = 0; k < 4; k++) { st[k] = v[k]; if (dl.stats_host) dl.stats_host[b * 8 + k] = v[k]; }
    }
}

__global__ void k_gather_counts(const int *__restrict__ countsL, const int *__restrict__ countsR, const int *__restrict__ stats,
                                int32_t *__restrict__ dst, int n_pairs)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_pairs) return;
    dst[3 * b + 0] = countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    dst[3 * b + 1] = countsR[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    dst[3 * b + 2] = stats[b * 8 + 3];
}

void launch_gather_counts(const int *countsL, const int *countsR, const int *stats, int32_t *dst, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_gather_counts, dim3((n_pairs + 255) / 256), dim3(256), 0, s, countsL, countsR, stats, dst, n_pairs);
}

void launch_stereo(const Geometry &g, const ImageSrc &srcL, const uint8_t *slabL, const ImageSrc &srcR, const uint8_t *slabR,
                   const int32_t *outL, const int *countsL, const uint8_t *descL,
                   const int32_t *outR, const int *countsR, const uint8_t *descR, const int *row_tabR,
                   float *u_right, float *depth, int *best_l1, unsigned *aux, StereoArgs a, int n_pairs, hipStream_t s, int *diag)
{
    // Passes of four left keypoints per wave.  More passes = fewer instructions per keypoint (phases A2 and C once per wave, fuller task lists in
    // phase B), but a wave runs its passes one after the other and each pass is a chain of dependent loads: the launch must still consist of
    // many more waves than the chip holds.  8 passes if that leaves >= 24 k waves (KAIST-shaped batches), else 6 / 4 / 2 with >= 12 k waves;
    // a single pair 1 - a frame waits for this kernel.  Measured, pairs/s of the whole pipeline on one box with 8 / 6 / 4 / 2 passes:
    // C2 123.7 / 123.7 / 122.5 / 121.7 k (round-4 kernel before this one: 121.0 k); k_stereo alone per step C2 101 / 103 / 102 / 112 us,
    // C3 (before the batched candidate loads) 105 / 98 / 97 / 101 us, C5 359 / 354 / 364 / 393 us.
    static const int env_pass = experiment_env("JSORB_STEREO_PASSES") ? std::max(1, std::min(ST_MAX_PASS, atoi(experiment_env("JSORB_STEREO_PASSES")))) : 0;
    auto waves = [&](int np) { return (long)n_pairs * ((g.T + SKPW * np - 1) / (SKPW * np)); };
    int npass = 1;
    if (env_pass) npass = env_pass;
    else if (n_pairs > 1) npass = waves(8) >= 24576 ? 8 : waves(6) >= 12288 ? 6 : waves(4) >= 12288 ? 4 : waves(2) >= 12288 ? 2 : 1;
    const int kpw = SKPW * npass;
    hipLaunchKernelGGL(k_stereo, xcd_grid((g.T + kpw - 1) / kpw, n_pairs), dim3(64), 0, s, g, srcL, slabL, srcR, slabR, outL, countsL, descL,
                       outR, countsR, descR, row_tabR, u_right, depth, best_l1, aux, a, n_pairs, diag, npass);
}

void launch_median(const Geometry &g, const int *countsL, float *u_right, float *depth, const int *best_l1, const unsigned *aux,
                   int *stats, int n_pairs, hipStream_t s, DeliverStereo dl)
{
    if (g.T <= 256 * 32) hipLaunchKernelGGL(k_median<32>, dim3(n_pairs), dim3(256), 0, s, g, countsL, u_right, depth, best_l1, aux, stats, dl);
    else hipLaunchKernelGGL(k_median<64>, dim3(n_pairs), dim3(256), 0, s, g, countsL, u_right, depth, best_l1, aux, stats, dl);
}

} // namespace jsorb
