// k_nms_ms.hip - multi-scale non-maximum suppression ("Pyramidal Feature Aggregation"), applied to the per-tile candidates
// between k_detect and k_compact when ORBextractor.apply_nms_ms is set (KITTI / KAIST / realsense yamls).
//
// Semantics restated:
//  GPU mode (nms_ms_mode_gpu = 1, what every shipped yaml selects): Fill_s0_score_kernel, NMS_S_s0_score_kernel,
//    NMS_L_s0_score_kernel  src/cuda/orb_FAST_apply_NMS_MS.cu:18-49, 235-310, 314-400 (launcher :402-467, orchestration
//    orb_gpu.cpp:667-696).  Every candidate is projected to level-0 coordinates (h, w) = trunc((y, x) * scale[level]).  For a
//    cell, sum = sum of the scores of all levels projecting there and zeros = number of levels that do not; a candidate survives
//    iff sum*zeros of its cell is >= sum*zeros of the 8 neighbouring cells.  The reference's NMS_S kernel zeroes its scatter
//    plane while other threads may still read it; the definition adopted (oracle and here) is "all reads
//    happen before any zeroing", which makes (sum, zeros) a function of the candidate set only.
//  CPU mode (nms_ms_mode_gpu = 0): ORB_GPU::FAST_apply_NMS_MS_cpu  src/cuda/orb_FAST_apply_NMS_MS.cpp:15-121: candidates are
//    binned by level-0 tile in level-major / tile-raster order and suppressed pairwise in that order (different levels, within
//    +-1 px, lower score dies, a tie kills the later one).
// MI355X design: the reference keeps an L x H0 x W0 int32 scatter volume (23 MB per 752x480 image) plus two H0 x W0 planes and
// clears one of them every frame; here ONE packed H0 x W0 accumulator per image (bits 0..23 sum, bits 24.. count) is updated
// with atomics by the few thousand candidates and cleaned by them afterwards, so no plane is ever streamed.  CPU mode sorts (bin << 16 | index) keys with an in-LDS bitonic
// sort to recover the reference's insertion order, then one thread replays the pairwise loop of each bin.
#include "jsorb_launch.h"

namespace jsorb {

__device__ __forceinline__ int level_of_tile(const Geometry &g, int idx)
{
    int lvl = 0;
#pragma unroll 1
    for (int i = 1; i < g.L; i++)
        if (idx >= g.lv[i].tile_off) lvl = i;
    return lvl;
}

#define MS_MARK (1ull << 63)

// GPU-mode semantics as three grid-wide passes (one launch each: the kernel boundary is the "all adds before any read, all reads
// before any zeroing" barrier).  A single workgroup per image with workgroup barriers between the passes was pure latency: 67 us
// per 32 images at 752x480, 304 us at 1280x720 - longer than the FAST kernel.
template <int PASS>
__global__ __launch_bounds__(256) void k_nms_ms_gpu(Geometry g, unsigned long long *tile_out, int *grid_all)
{
    const int idx = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (idx >= g.T) return;
    const int H0 = g.lv[0].H, W0 = g.lv[0].W, L = g.L;
    int *grid = grid_all + (size_t)b * H0 * W0;
    unsigned long long *t = tile_out + (size_t)b * g.T;
    const unsigned long long p = t[idx];
    const int s = kp_score(p);
    if (!s) return;
    const float sc = g.lv[level_of_tile(g, idx)].scale;
    const int h = (int)((float)kp_y(p) * sc), w = (int)((float)kp_x(p) * sc);
    if (PASS == 0) {
        // pass A: scatter-add (score | 1<<24) into the level-0 accumulator
        atomicAdd(&grid[(size_t)h * W0 + w], s | (1 << 24));
    } else if (PASS == 1) {
        // pass B: compare sum*zeros with the 3x3 neighbourhood
        const int c = grid[(size_t)h * W0 + w];
        const int mine = (c & 0xFFFFFF) * (L - (c >> 24));
        bool valid = true;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int hh = h + dy, ww = w + dx;
                int nb = 0;
                if (hh >= 0 && hh < H0 && ww >= 0 && ww < W0) {
                    const int q = grid[(size_t)hh * W0 + ww];
                    nb = (q & 0xFFFFFF) * (L - (q >> 24));
                }
                valid = valid && (mine >= nb);
            }
        if (!valid) t[idx] = p | MS_MARK;
    } else {
        // pass C: clean the accumulator (it stays all-zero between frames) and apply the verdicts
        grid[(size_t)h * W0 + w] = 0;
        if (p & MS_MARK) t[idx] = p & ~(MS_MARK | (0xFFFull << 32));
    }
}

// ---- CPU-mode semantics --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l0_coords(const Geometry &g, unsigned long long p, int lvl, int &x_l0, int &y_l0)
{
    const float sc = g.lv[lvl].scale;
    x_l0 = (int)((float)kp_x(p) * sc - (float)JSORB_BORDER);
    y_l0 = (int)((float)kp_y(p) * sc - (float)JSORB_BORDER);
}

__global__ __launch_bounds__(1024) void k_nms_ms_cpu(Geometry g, unsigned long long *tile_out, int *scratch_all, int n_pad)
{
    extern __shared__ unsigned s_key[];
    const int tid = threadIdx.x, b = blockIdx.x;
    unsigned long long *t = tile_out + (size_t)b * g.T;
    int *score = scratch_all + (size_t)b * g.T;           // mutable copy of the scores (nms_ms_cpu_score_)
    const int th0 = g.lv[0].th, tw0 = g.lv[0].tw, ntw0 = g.lv[0].ntw;
    for (int idx = tid; idx < n_pad; idx += 1024) {
        unsigned key = 0xFFFFFFFFu;
        if (idx < g.T) {
            const unsigned long long p = t[idx];
            const int s = kp_score(p);
            score[idx] = s;
            if (s > 0) {
                int x_l0, y_l0;
                l0_coords(g, p, level_of_tile(g, idx), x_l0, y_l0);
                key = ((unsigned)((y_l0 / th0) * ntw0 + x_l0 / tw0) << 16) | (unsigned)idx;
            }
        }
        s_key[idx] = key;
    }
    __syncthreads();
    // bitonic sort, ascending: (bin, index) order == the reference's push_back order within each bin
    for (int k = 2; k <= n_pad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pad; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned a = s_key[i], c = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { s_key[i] = c; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // one thread per bin replays the pairwise suppression loop (orb_FAST_apply_NMS_MS.cpp:70-103)
    for (int i = tid; i < n_pad; i += 1024) {
        const unsigned key = s_key[i];
        if (key == 0xFFFFFFFFu) continue;
        const unsigned bin = key >> 16;
        if (i > 0 && (s_key[i - 1] >> 16) == bin) continue;             // not the first entry of its bin
        int n = 1;
        while (i + n < n_pad && (s_key[i + n] >> 16) == bin && s_key[i + n] != 0xFFFFFFFFu) n++;
        for (int j = 0; j < n; j++)
            for (int k = 0; k < n; k++) {
                if (j == k) continue;
                const int ij = (int)(s_key[i + j] & 0xFFFFu), ik = (int)(s_key[i + k] & 0xFFFFu);
                const int lj = level_of_tile(g, ij), lk = level_of_tile(g, ik);
                if (lj == lk) continue;
                const int sj = score[ij], sk = score[ik];
                if (sj && sk) {
                    int xj, yj, xk, yk;
                    l0_coords(g, t[ij], lj, xj, yj);
                    l0_coords(g, t[ik], lk, xk, yk);
                    const int xd = xj - xk, yd = yj - yk;
                    if (xd >= -1 && xd <= 1 && yd >= -1 && yd <= 1) {
                        if (sj < sk) score[ij] = 0;
                        else score[ik] = 0;
                    }
                }
            }
        for (int j = 0; j < n; j++) {
            const int ij = (int)(s_key[i + j] & 0xFFFFu);
            if (score[ij] == 0) t[ij] &= ~(0xFFFull << 32);
        }
    }
}

void launch_nms_ms(const Geometry &g, unsigned long long *tile_out, int *ms_grid, int *ms_scratch, int mode_gpu, int n_images, hipStream_t s)
{
    if (mode_gpu) {
        const dim3 grid((g.T + 255) / 256, n_images);
        hipLaunchKernelGGL(k_nms_ms_gpu<0>, grid, dim3(256), 0, s, g, tile_out, ms_grid);
        hipLaunchKernelGGL(k_nms_ms_gpu<1>, grid, dim3(256), 0, s, g, tile_out, ms_grid);
        hipLaunchKernelGGL(k_nms_ms_gpu<2>, grid, dim3(256), 0, s, g, tile_out, ms_grid);
    } else {
        int n_pad = 1024;
        while (n_pad < g.T) n_pad <<= 1;
        hipLaunchKernelGGL(k_nms_ms_cpu, dim3(n_images), dim3(1024), (size_t)n_pad * 4, s, g, tile_out, ms_scratch, n_pad);
    }
}

} // namespace jsorb
