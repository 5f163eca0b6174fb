// k_compact.hip - device-side, order-preserving compaction of the per-tile candidates.
//
// Replaces the reference's pipeline bubble FAST_obtain_keypoints() (src/cuda/orb_FAST_obtain_keypoints.cpp:12-56:
// full device sync, D2H of 3T ints, CPU loop, H2D of 3T ints) by one workgroup per image: a wave64
// ballot/popcount prefix inside each wave and a 16-entry LDS scan across the waves.  Output order is identical
// to the CPU loop: level-major, tile-raster within a level, candidates with score > 0 only.
// Besides the compacted list it emits per-level counts (n_keypoints_[i]) and, for the stereo matcher, the index
// of the first keypoint of every tile row (keypoints of one tile row are contiguous in the output) and of every tile, and
// (flat form) the keypoints counting-sorted by (level, level-0 row): the matcher's scan-line buckets.
//
// Scan-line buckets.  The reference builds vRowIndices on the CPU: right keypoint iR is listed under every row of
// [floor(y - r), ceil(y + r)], r = 2 * scale[octave] (orb_stereo_match.cu:119-140), and a left keypoint looks at the list of its
// row.  Here every keypoint is listed ONCE, under (level, its level-0 row): the keypoints of level l that cover row v have their
// row in (v - 1 - r_l, v + 1 + r_l), a contiguous run of buckets of that level, which k_stereo reads off the start table and
// filters with the reference's exact tests.  Two LDS atomics per keypoint instead of one per covered row; the order inside a
// bucket is whatever the atomics give - the matcher's (distance, iR) min-key does not depend on it.
#include "jsorb_launch.h"
#include "k_compact_body.h"

namespace jsorb {

__global__ __launch_bounds__(1024) void k_compact(Geometry g, const unsigned long long *__restrict__ tile_out,
                                                  unsigned long long *__restrict__ kp, int *__restrict__ counts,
                                                  int *__restrict__ row_tab, int *__restrict__ counts_host)
{
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int b = blockIdx.x;
    const unsigned long long *tin = tile_out + (size_t)b * g.T;
    unsigned long long *kout = kp + (size_t)b * g.T;
    int *rt = row_tab + (size_t)b * g.row_tab_stride;
    int *tp = rt + g.row_tab_len;                              // per-tile start table: index of the first keypoint at or after tile j (T + 1 entries)
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
            if (j < n) tp[lv.tile_off + j] = pos;
            if (j < n && (j % lv.ntw) == 0) rt[lv.row_tab_off + j / lv.ntw] = pos;
            base += tot;
            __syncthreads();
        }
        if (tid == 0) {
            rt[lv.row_tab_off + lv.nth] = base;
            counts[b * (JSORB_MAX_LEVELS + 1) + lvl] = base - level_start;
            if (counts_host) counts_host[b * (JSORB_MAX_LEVELS + 1) + lvl] = base - level_start;
        }
    }
    if (tid == 0) {
        tp[g.T] = base;
        counts[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS] = base;
        if (counts_host) counts_host[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS] = base;
    }
}

template <int NC, int NT>
__global__ __launch_bounds__(NT) void k_compact_flat(Geometry g, const unsigned long long *__restrict__ tile_out,
                                                     unsigned long long *__restrict__ kp, int *__restrict__ counts,
                                                     int *__restrict__ row_tab, int *__restrict__ counts_host)
{
    extern __shared__ int s_epi_dyn[];                          // L * epi_rows bucket counters (scan-line buckets), then cursors
    compact_flat_workgroup<NC, NT, CMP_MAX_CHUNKS * 16>(g, tile_out, kp, counts, row_tab, counts_host, (int)blockIdx.x, s_epi_dyn);
}

// threads of k_compact_flat's workgroup on batch handles (see the kernel)
#ifndef CMP_NT_BATCH
#define CMP_NT_BATCH 256
#endif

void launch_compact(const Geometry &g, const unsigned long long *tile_out, unsigned long long *kp, int *counts,
                    int *row_tab, int n_images, hipStream_t s, int *counts_host)
{
    const size_t epi = g.epi_rows ? (size_t)g.L * g.epi_rows * sizeof(int) : 0;
    constexpr int NB = CMP_NT_BATCH;
    // single images (latency layouts): 1024 threads, the kernel is on the frame's critical path and has the chip to itself; batches: small workgroups
    // (images with more than 4096 tiles keep 1024 threads: the re-reading form with a quarter of the threads took 0.18 instead of 0.07 ms per step at the
    // KAIST shape and cost more than the starvation it avoids - 35.4 k against 35.7 k pairs/s)
    if (g.T <= 4 * 1024 && g.latency)
        hipLaunchKernelGGL((k_compact_flat<4, 1024>), dim3(n_images), dim3(1024), epi, s, g, tile_out, kp, counts, row_tab, counts_host);
    else if (g.T <= 4 * 1024)
        hipLaunchKernelGGL((k_compact_flat<4 * 1024 / NB, NB>), dim3(n_images), dim3(NB), epi, s, g, tile_out, kp, counts, row_tab, counts_host);
    // (4096 < T <= 8192 - the KITTI-shaped images - in the register form with 512 threads: measured, no difference)
    else if (g.T <= CMP_MID_T && !g.latency)
        hipLaunchKernelGGL((k_compact_flat<0, NB>), dim3(n_images), dim3(NB), epi, s, g, tile_out, kp, counts, row_tab, counts_host);
    else if (g.T <= CMP_MAX_CHUNKS * 1024)
        hipLaunchKernelGGL((k_compact_flat<0, 1024>), dim3(n_images), dim3(1024), epi, s, g, tile_out, kp, counts, row_tab, counts_host);
    else
        hipLaunchKernelGGL(k_compact, dim3(n_images), dim3(1024), 0, s, g, tile_out, kp, counts, row_tab, counts_host);
}

} // namespace jsorb
