// k_stereo.hip - stereo matcher on device: row-epipolar candidate search + Hamming brute force + 11-shift 11x11 L1
// refinement + sub-pixel parabola + depth, one wave64 per left keypoint, all pairs of a batch in ONE launch;
// then one workgroup per pair for the 2.1 x median outlier cut.
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
// Candidate pruning uses the tile-row start table produced by k_compact (keypoints of one tile row are contiguous
// and ordered), then applies the reference's exact row / octave / u tests; candidate order (ascending iR) only matters
// for ties, which the (distance << 20 | iR) min-key reproduces.
#include "jsorb_launch.h"

namespace jsorb {

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(256) void k_stereo(Geometry g, ImageSrc srcL, const uint8_t *slabL, ImageSrc srcR, const uint8_t *slabR,
                                                const int32_t *__restrict__ outL, const int *__restrict__ countsL, const uint8_t *__restrict__ descL,
                                                const int32_t *__restrict__ outR, const int *__restrict__ countsR, const uint8_t *__restrict__ descR,
                                                const int *__restrict__ row_tabR,
                                                float *__restrict__ u_right, float *__restrict__ depth, int *__restrict__ best_l1,
                                                int *__restrict__ stats, StereoArgs sa)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + wave;
    const int Nl = countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    const int Nr = countsR[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    if (i >= Nl) return;
    const int32_t *oL = outL + (size_t)b * 6 * g.T;
    const int32_t *oR = outR + (size_t)b * 6 * g.T;
    const int xL0 = oL[i], yL0 = oL[Nl + i], levelL = oL[4 * (size_t)Nl + i];
    const float uL = (float)xL0, vL = (float)yL0;
    const float minU = uL - sa.maxD, maxU = uL - 0.0f;
    const size_t tb = (size_t)b * g.T;

    unsigned best_key = 0xFFFFFFFFu;
    int n_cand = 0;
    if (!(maxU < 0)) {
        const uint4 *dl = reinterpret_cast<const uint4 *>(descL + (tb + i) * 32);
        const uint4 a0 = dl[0], a1 = dl[1];
        const int vLi = (int)vL;
        const int *rt = row_tabR + (size_t)b * g.row_tab_len;
        for (int lr = levelL - 1; lr <= levelL + 1; lr++) {
            if (lr < 0 || lr >= g.L) continue;
            const LevelDesc &lv = g.lv[lr];
            const float r = 2.0f * lv.scale;
            // conservative tile-row window of level lr (exact tests follow per candidate)
            int lo = (int)__builtin_floorf((vL - r - 1.0f) / lv.scale) - 1;
            int hi = (int)__builtin_ceilf((vL + r + 2.0f) / lv.scale) + 1;
            int t_lo = lo < 0 ? 0 : lo / lv.th;
            int t_hi = hi < 0 ? -1 : hi / lv.th;
            if (t_hi > lv.nth - 1) t_hi = lv.nth - 1;
            if (t_lo > t_hi) continue;
            const int j0 = rt[lv.row_tab_off + t_lo], j1 = rt[lv.row_tab_off + t_hi + 1];
            for (int j = j0 + lane; j < j1; j += 64) {
                const float kpY = (float)oR[Nr + j];
                const int maxr = (int)__builtin_ceilf(kpY + r), minr = (int)__builtin_floorf(kpY - r);
                if (vLi < minr || vLi > maxr) continue;
                const float uR = (float)oR[j];
                if (!(uR >= minU && uR <= maxU)) continue;
                n_cand++;
                const uint4 *dr = reinterpret_cast<const uint4 *>(descR + (tb + j) * 32);
                const int d = hamming256(a0, a1, dr[0], dr[1]);
                if (d < sa.th_high) {
                    const unsigned key = ((unsigned)d << 20) | (unsigned)j;
                    best_key = key < best_key ? key : best_key;
                }
            }
        }
    }
    best_key = wave_min_u32(best_key);
    n_cand = wave_sum_i32(n_cand);

    float out_u = -1.0f, out_d = -1.0f;
    int out_l1 = -1, corr = 0;
    if (best_key != 0xFFFFFFFFu && (int)(best_key >> 20) < sa.th_orb) {
        const int bestIdxR = (int)(best_key & 0xFFFFFu);
        const LevelDesc &lv = g.lv[levelL];
        const float uR0 = (float)oR[bestIdxR];
        const float scaleFactor = lv.inv_scale;
        const float scaleduR0 = roundf(uR0 * scaleFactor);
        const float scaleduL0 = roundf(uL * scaleFactor);
        const float scaledvL0 = roundf(vL * scaleFactor);
        const float iniu = scaleduR0 - 5.0f - 5.0f, endu = scaleduR0 + 5.0f + 5.0f;
        if (!(iniu < 0 || endu >= (float)lv.W)) {
            corr = 1;
            const int xl = (int)scaleduL0, xr = (int)scaleduR0, y = (int)scaledvL0;
            int pl, pr;
            const uint8_t *li = level_ptr(g, srcL, slabL, b, levelL, pl) + (size_t)y * pl + xl;
            const uint8_t *ri = level_ptr(g, srcR, slabR, b, levelL, pr) + (size_t)y * pr + xr;
            const int lc = li[0];
            int acc[11];
#pragma unroll
            for (int s = 0; s < 11; s++) acc[s] = 0;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                const int idx = pass * 64 + lane;
                if (idx < 121) {
                    const int wh = idx / 11 - 5, ww = idx % 11 - 5;
                    const int lval = (int)li[wh * pl + ww] - lc;
                    const uint8_t *rrow = ri + wh * pr + ww;
#pragma unroll
                    for (int s = 0; s < 11; s++) {
                        const int rval = (int)rrow[s - 5] - (int)ri[s - 5];
                        const int df = lval - rval;
                        acc[s] += df < 0 ? -df : df;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 11; s++) acc[s] = wave_sum_i32(acc[s]);
            int bestDist = 0x7FFFFFFF, bestR = 0;
#pragma unroll
            for (int s = 0; s < 11; s++)
                if (acc[s] < bestDist) { bestDist = acc[s]; bestR = s; }
            if (!(bestR == 0 || bestR == 10)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int s = 1; s < 10; s++)
                    if (s == bestR) { dist1 = (float)acc[s - 1]; dist2 = (float)acc[s]; dist3 = (float)acc[s + 1]; }
                const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = lv.scale * ((scaleduR0 + (float)bestR - 5.0f) + deltaR);
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
    }
    if (lane == 0) {
        u_right[tb + i] = out_u;
        depth[tb + i] = out_d;
        best_l1[tb + i] = out_l1;
        int *st = stats + b * 8;
        if (n_cand) atomicAdd(&st[0], n_cand);
        if (corr) atomicAdd(&st[1], 1);
        if (out_l1 >= 0) atomicAdd(&st[2], 1);
    }
}

// 2.1 x median cut (orb_stereo_match.cu:563-578).  The median of the sorted (dist, idx) pairs is the (nv/2)-th smallest
// distance; L1 distances are < 2^16 (121*510), so a two-pass 256-bin radix select in LDS finds it exactly.
__global__ __launch_bounds__(256) void k_median(Geometry g, const int *__restrict__ countsL, float *__restrict__ u_right,
                                                float *__restrict__ depth, const int *__restrict__ best_l1, int *__restrict__ stats)
{
    __shared__ int hist[256];
    __shared__ int sel[4];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int Nl = countsL[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS];
    const size_t tb = (size_t)b * g.T;
    int *st = stats + b * 8;
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < Nl; i += 256) {
        const int d = best_l1[tb + i];
        if (d >= 0) atomicAdd(&hist[(d >> 8) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int nv = 0;
        for (int k = 0; k < 256; k++) nv += hist[k];
        sel[2] = nv;
        int kth = nv / 2, cum = 0, bin = 0;
        for (int k = 0; k < 256; k++) {
            if (cum + hist[k] > kth) { bin = k; break; }
            cum += hist[k];
        }
        sel[0] = bin;
        sel[1] = kth - cum;     // rank inside the bin
    }
    __syncthreads();
    const int nv = sel[2];
    if (nv == 0) {              // Appendix C-6: nothing matched -> no cut
        if (tid == 0) st[3] = 0;
        return;
    }
    const int bin = sel[0], kin = sel[1];
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < Nl; i += 256) {
        const int d = best_l1[tb + i];
        if (d >= 0 && ((d >> 8) & 255) == bin) atomicAdd(&hist[d & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int cum = 0, low = 0;
        for (int k = 0; k < 256; k++) {
            if (cum + hist[k] > kin) { low = k; break; }
            cum += hist[k];
        }
        sel[3] = (bin << 8) | low;
        st[3] = nv;
    }
    __syncthreads();
    const float median = (float)sel[3];
    const float thDist = 1.5f * 1.4f * median;
    int removed = 0;
    for (int i = tid; i < Nl; i += 256) {
        const int d = best_l1[tb + i];
        if (d >= 0 && !((float)d < thDist)) {
            u_right[tb + i] = -1.0f;
            depth[tb + i] = -1.0f;
            removed++;
        }
    }
    if (removed) atomicSub(&st[3], removed);
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
                   float *u_right, float *depth, int *best_l1, int *stats, StereoArgs a, int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_stereo, dim3((g.T + 3) / 4, n_pairs), dim3(256), 0, s, g, srcL, slabL, srcR, slabR, outL, countsL, descL,
                       outR, countsR, descR, row_tabR, u_right, depth, best_l1, stats, a);
}

void launch_median(const Geometry &g, const int *countsL, float *u_right, float *depth, const int *best_l1, int *stats,
                   int n_pairs, hipStream_t s)
{
    hipLaunchKernelGGL(k_median, dim3(n_pairs), dim3(256), 0, s, g, countsL, u_right, depth, best_l1, stats);
}

} // namespace jsorb
