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

// Flat form for T <= 65536 candidates: one pass stores the wave ballots of all 1024-entry chunks, one workgroup prefix scan turns
// them into start positions, a second pass writes; positions of arbitrary candidates (tile-row starts, level boundaries) are then
// read off the same tables.  Three barriers instead of two per level and chunk - the kernel is one workgroup per image and pure
// latency.
#define CMP_MAX_CHUNKS 64
__device__ __forceinline__ int compact_pos_of(int j, int T, int total, const unsigned long long *s_bal, const int *s_base)
{
    if (j >= T) return total;
    const int cell = j >> 6;                                  // chunk * 16 + wave
    return s_base[cell] + __popcll(s_bal[cell] & ((1ull << (j & 63)) - 1ull));
}

// NC > 0: the image has at most NC chunks of 1024 tiles and a thread keeps its NC candidates in registers over the three passes - one
// memory round trip for all of them instead of one per chunk and pass, and the bucket scatter at the end works from the registers instead of
// re-reading the list the workgroup has just written (single frames: 10 -> 6 us of a kernel every other kernel of the frame waits for).
template <int NC>
__global__ __launch_bounds__(1024) void k_compact_flat(Geometry g, const unsigned long long *__restrict__ tile_out,
                                                       unsigned long long *__restrict__ kp, int *__restrict__ counts,
                                                       int *__restrict__ row_tab, int *__restrict__ counts_host)
{
    __shared__ unsigned long long s_bal[CMP_MAX_CHUNKS * 16];
    __shared__ int s_base[CMP_MAX_CHUNKS * 16];
    __shared__ int s_wtot[16];
    __shared__ int s_total;
    __shared__ float s_scale[JSORB_MAX_LEVELS];
    __shared__ int s_toff[JSORB_MAX_LEVELS];                   // first tile of every level (a loop over the kernel arguments paid a scalar-load round trip per level and candidate)
    extern __shared__ int s_epi[];                             // L * epi_rows bucket counters (scan-line buckets), then cursors
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int b = blockIdx.x, T = g.T;
    const int EH = g.epi_rows, EN = g.L * EH;
    for (int t = tid; t < EN; t += 1024) s_epi[t] = 0;
    if (tid < JSORB_MAX_LEVELS) {
        s_scale[tid] = tid < g.L ? g.lv[tid].scale : 0.0f;
        s_toff[tid] = tid < g.L ? g.lv[tid].tile_off : 0x7FFFFFFF;
    }
    const unsigned long long *tin = tile_out + (size_t)b * T;
    unsigned long long *kout = kp + (size_t)b * T;
    int *rt = row_tab + (size_t)b * g.row_tab_stride;
    int *tp = rt + g.row_tab_len;                              // per-tile start table: index of the first keypoint at or after tile j (T + 1 entries)
    const int n_chunks = (T + 1023) >> 10, n_cells = n_chunks * 16;
    constexpr int NR = NC > 0 ? NC : 1;
    unsigned long long preg[NR];
    int posr[NR], bktr[NR];
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NR; c++) {
            const int j = c * 1024 + tid;
            preg[c] = j < T ? tin[j] : 0ull;
        }
#pragma unroll
        for (int c = 0; c < NR; c++) {
            if (c >= n_chunks) break;
            const unsigned long long bal = __ballot(kp_score(preg[c]) > 0);
            if (lane == 0) s_bal[c * 16 + wave] = bal;
        }
    } else {
        for (int c = 0; c < n_chunks; c++) {
            const int j = c * 1024 + tid;
            const unsigned long long p = j < T ? tin[j] : 0ull;
            const unsigned long long bal = __ballot(kp_score(p) > 0);
            if (lane == 0) s_bal[c * 16 + wave] = bal;
        }
    }
    __syncthreads();
    {   // exclusive prefix over the n_cells <= 1024 cells (one per thread)
        const int v = tid < n_cells ? __popcll(s_bal[tid]) : 0;
        const int incl = wave_inclusive_scan_i32(v);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int t = s_wtot[w];
            if (w < wave) base += t;
            tot += t;
        }
        if (tid < n_cells) s_base[tid] = base + incl - v;
        if (tid == 0) s_total = tot;
    }
    __syncthreads();
    const int total = s_total;
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NR; c++) {
            const int j = c * 1024 + tid;
            posr[c] = -1; bktr[c] = 0;
            if (j < T && kp_score(preg[c]) > 0) {
                int lvl = 0;
#pragma unroll
                for (int i = 1; i < JSORB_MAX_LEVELS; i++) lvl += j >= s_toff[i] ? 1 : 0;
                preg[c] |= (unsigned long long)lvl << 44;
                posr[c] = compact_pos_of(j, T, total, s_bal, s_base);
                kout[posr[c]] = preg[c];
                bktr[c] = lvl * EH + min((int)((float)kp_y(preg[c]) * s_scale[lvl]), EH - 1);
                if (EN) atomicAdd(&s_epi[bktr[c]], 1);
            }
        }
    } else {
        for (int c = 0; c < n_chunks; c++) {
            const int j = c * 1024 + tid;
            if (j >= T) break;
            const unsigned long long p = tin[j];
            if (kp_score(p) > 0) {
                int lvl = 0;
#pragma unroll
                for (int i = 1; i < JSORB_MAX_LEVELS; i++) lvl += j >= s_toff[i] ? 1 : 0;
                kout[compact_pos_of(j, T, total, s_bal, s_base)] = p | ((unsigned long long)lvl << 44);
                if (EN) atomicAdd(&s_epi[lvl * EH + min((int)((float)kp_y(p) * s_scale[lvl]), EH - 1)], 1);
            }
        }
    }
    // first keypoint at or after every tile (the stereo matcher's column pruning)
    for (int t = tid; t <= T; t += 1024) tp[t] = compact_pos_of(t, T, total, s_bal, s_base);
    // first keypoint of every tile row (+ the end of each level), per-level counts
    for (int t = tid; t < g.row_tab_len; t += 1024) {
        int lvl = 0;
#pragma unroll 1
        for (int i = 1; i < g.L; i++)
            if (t >= g.lv[i].row_tab_off) lvl = i;
        const LevelDesc &lv = g.lv[lvl];
        const int k = t - lv.row_tab_off;                     // 0 .. nth
        rt[t] = compact_pos_of(lv.tile_off + k * lv.ntw, T, total, s_bal, s_base);
    }
    if (tid < g.L) {
        const int j0 = g.lv[tid].tile_off, j1 = tid + 1 < g.L ? g.lv[tid + 1].tile_off : T;
        const int c = compact_pos_of(j1, T, total, s_bal, s_base) - compact_pos_of(j0, T, total, s_bal, s_base);
        counts[b * (JSORB_MAX_LEVELS + 1) + tid] = c;
        if (counts_host) counts_host[b * (JSORB_MAX_LEVELS + 1) + tid] = c;
    }
    if (tid == 0) {
        counts[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS] = total;
        if (counts_host) counts_host[b * (JSORB_MAX_LEVELS + 1) + JSORB_MAX_LEVELS] = total;
    }
    if (!EN) return;
    // ---- scan-line buckets: exclusive scan of the bucket counts (a contiguous run per thread), then the scatter ----
    int *et = rt + g.epi_off;                                  // EN + 1 bucket starts, then the entries
    int2 *ee = reinterpret_cast<int2 *>(et + ((EN + 2) & ~1));
    __syncthreads();
    {
        const int per = (EN + 1023) >> 10, t0 = tid * per, t1 = min(t0 + per, EN);
        int sum = 0;
        for (int t = t0; t < t1; t++) sum += s_epi[t];
        const int incl = wave_inclusive_scan_i32(sum);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < 16; w++)
            if (w < wave) base += s_wtot[w];
        int run = base + incl - sum;
        for (int t = t0; t < t1; t++) {
            const int c = s_epi[t];
            s_epi[t] = run;
            et[t] = run;
            run += c;
        }
        if (tid == 0) et[EN] = total;
    }
    __syncthreads();
    // dense pass over the compacted list this workgroup has just written (level in the record, scale from LDS): four independent
    // records per thread and round, so that their load latencies overlap - the kernel is one workgroup per image and pure latency
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NR; c++) {
            if (posr[c] < 0) continue;
            const int lvl = kp_level(preg[c]);
            const float sc = s_scale[lvl];
            const int yi = (int)((float)kp_y(preg[c]) * sc), xi = (int)((float)kp_x(preg[c]) * sc);
            const int slot = atomicAdd(&s_epi[bktr[c]], 1);
            ee[slot] = make_int2(posr[c], (xi & 0xFFFF) | (yi << 16));
        }
        return;
    }
    for (int i0 = tid; i0 < total; i0 += 4096) {
        unsigned long long p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = i0 + 1024 * u < total ? kout[i0 + 1024 * u] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + 1024 * u >= total) break;
            const int lvl = kp_level(p[u]);
            const float sc = s_scale[lvl];
            const int yi = (int)((float)kp_y(p[u]) * sc), xi = (int)((float)kp_x(p[u]) * sc);      // the level-0 coordinates k_describe packs (K11)
            const int slot = atomicAdd(&s_epi[lvl * EH + min(yi, EH - 1)], 1);
            ee[slot] = make_int2(i0 + 1024 * u, (xi & 0xFFFF) | (yi << 16));
        }
    }
}

void launch_compact(const Geometry &g, const unsigned long long *tile_out, unsigned long long *kp, int *counts,
                    int *row_tab, int n_images, hipStream_t s, int *counts_host)
{
    if (g.T <= 4 * 1024)
        hipLaunchKernelGGL(k_compact_flat<4>, dim3(n_images), dim3(1024), g.epi_rows ? (size_t)g.L * g.epi_rows * sizeof(int) : 0, s, g, tile_out, kp, counts,
                           row_tab, counts_host);
    else if (g.T <= CMP_MAX_CHUNKS * 1024)
        hipLaunchKernelGGL(k_compact_flat<0>, dim3(n_images), dim3(1024), g.epi_rows ? (size_t)g.L * g.epi_rows * sizeof(int) : 0, s, g, tile_out, kp, counts,
                           row_tab, counts_host);
    else
        hipLaunchKernelGGL(k_compact, dim3(n_images), dim3(1024), 0, s, g, tile_out, kp, counts, row_tab, counts_host);
}

} // namespace jsorb
