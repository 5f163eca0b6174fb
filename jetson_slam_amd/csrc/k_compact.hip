// k_compact.hip - device-side, order-preserving compaction of the per-tile candidates.
//
// Replaces the reference's pipeline bubble FAST_obtain_keypoints() (src/cuda/orb_FAST_obtain_keypoints.cpp:12-56:
// full device sync, D2H of 3T ints, CPU loop, H2D of 3T ints) by one workgroup per image: a wave64
// ballot/popcount prefix inside each wave and a 16-entry LDS scan across the waves.  Output order is identical
// to the CPU loop: level-major, tile-raster within a level, candidates with score > 0 only.
// Besides the compacted list it emits per-level counts (n_keypoints_[i]) and, for the stereo matcher, the index
// of the first keypoint of every tile row (keypoints of one tile row are contiguous in the output).
#include "jsorb_launch.h"

namespace jsorb {

__global__ __launch_bounds__(1024) void k_compact(Geometry g, const unsigned long long *__restrict__ tile_out,
                                                  unsigned long long *__restrict__ kp, int *__restrict__ counts,
                                                  int *__restrict__ row_tab)
{
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const unsigned long long *tin = tile_out + (size_t)b * g.T;
    unsigned long long *kout = kp + (size_t)b * g.T;
    int *rt = row_tab + (size_t)b * g.row_tab_len;
    int base = 0;
    for (int lvl = 0; lvl < g.L; lvl++) {
        const LevelDesc &lv = g.lv[lvl];
        const int n = lv.nth * lv.ntw;
        const int level_start = base;
        for (int c0 = 0; c0 < n; c0 += 1024) {
            const int j = c0 + tid;
            unsigned long long p = 0;
            if (j < n) p = tin[lv.tile_off + j];
            const bool flag = kp_score(p) > 0;
            const unsigned long long bal = __ballot(flag);
            const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int wbase = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) {
                const int t = wave_tot[w];
                if (w < wave) wbase += t;
                tot += t;
            }
            const int pos = base + wbase + wpre;
            if (flag) kout[pos] = p | ((unsigned long long)lvl << 44);
            if (j < n && (j % lv.ntw) == 0) rt[lv.row_tab_off + j / lv.ntw] = pos;
            base += tot;
            __syncthreads();
        }
        if (tid == 0) {
            rt[lv.row_tab_off + lv.nth] = base;
            counts[b * (JSORB_MAX_LEVELS + 1) + lvl] = base - level_start;
        }
    }
    if (tid == 0) counts[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS] = base;
}

void launch_compact(const Geometry &g, const unsigned long long *tile_out, unsigned long long *kp, int *counts,
                    int *row_tab, int n_images, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact, dim3(n_images), dim3(1024), 0, s, g, tile_out, kp, counts, row_tab);
}

} // namespace jsorb
