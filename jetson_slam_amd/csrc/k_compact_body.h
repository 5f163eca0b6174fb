// k_compact_body.h - device side of the flat order-preserving compaction (see k_compact.hip for the design): the work of ONE workgroup = one image.
#pragma once
#include "jsorb_launch.h"

namespace jsorb {

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

// NC > 0: the image has at most NC chunks of NT tiles and a thread keeps its NC candidates in registers over the three passes - one
// memory round trip for all of them instead of one per chunk and pass, and the bucket scatter at the end works from the registers instead of
// re-reading the list the workgroup has just written (single frames: 10 -> 6 us of a kernel every other kernel of the frame waits for).
// NT: threads of the workgroup.  Round 6: inside the 4-lane batch pipeline a 1024-thread workgroup waits until ONE CU has 16 free wave slots, while the
// other lanes' kernels keep every CU full - a launch that takes 10 us alone took 45-70 us there and cost the step 2.5 x its stand-alone time
// (tools/micro/r6_skip.py); batch handles therefore run it with CMP_NT_BATCH threads and 1024 / CMP_NT_BATCH x the candidates per thread.
// MAXCELLS: cells of 64 candidates the workgroup's tables hold (T <= 64 * MAXCELLS).  b: the image; s_epi: L * epi_rows ints of (dynamic) LDS.
// A device function: the body of the kernels of its own (k_compact.hip) and of the compaction workgroups of the fused k_blur_compact launch (k_blur.hip).
template <int NC, int NT, int MAXCELLS>
__device__ __forceinline__ void compact_flat_workgroup(const Geometry &g, const unsigned long long *__restrict__ tile_out,
                                                       unsigned long long *__restrict__ kp, int *__restrict__ counts,
                                                       int *__restrict__ row_tab, int *__restrict__ counts_host, const int b, int *s_epi)
{
    constexpr int NW = NT / 64;                                // waves of the workgroup
    __shared__ unsigned long long s_bal[MAXCELLS];             // one ballot per cell of 64 consecutive candidates
    __shared__ int s_base[MAXCELLS];
    __shared__ int s_wtot[NW];
    __shared__ int s_total;
    __shared__ float s_scale[JSORB_MAX_LEVELS];
    __shared__ int s_toff[JSORB_MAX_LEVELS];                   // first tile of every level (a loop over the kernel arguments paid a scalar-load round trip per level and candidate)
    // (s_epi: L * epi_rows bucket counters (scan-line buckets), then cursors)
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);
    const int T = g.T;
    const int EH = g.epi_rows, EN = g.L * EH;
    for (int t = tid; t < EN; t += NT) s_epi[t] = 0;
    if (tid < JSORB_MAX_LEVELS) {
        s_scale[tid] = tid < g.L ? g.lv[tid].scale : 0.0f;
        s_toff[tid] = tid < g.L ? g.lv[tid].tile_off : 0x7FFFFFFF;
    }
    const unsigned long long *tin = tile_out + (size_t)b * T;
    unsigned long long *kout = kp + (size_t)b * T;
    int *rt = row_tab + (size_t)b * g.row_tab_stride;
    int *tp = rt + g.row_tab_len;                              // per-tile start table: index of the first keypoint at or after tile j (T + 1 entries)
    const int n_chunks = (T + NT - 1) / NT, n_cells = (T + 63) >> 6;      // chunk c of a thread: candidate c * NT + tid, in cell (c * NT + tid) >> 6 = c * NW + wave
    constexpr int NR = NC > 0 ? NC : 1;
    unsigned long long preg[NR];
    int posr[NR], bktr[NR];
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NR; c++) {
            const int j = c * NT + tid;
            preg[c] = j < T ? tin[j] : 0ull;
        }
#pragma unroll
        for (int c = 0; c < NR; c++) {
            if (c >= n_chunks) break;
            const unsigned long long bal = __ballot(kp_score(preg[c]) > 0);
            if (lane == 0) s_bal[c * NW + wave] = bal;
        }
    } else {
        for (int c = 0; c < n_chunks; c++) {
            const int j = c * NT + tid;
            const unsigned long long p = j < T ? tin[j] : 0ull;
            const unsigned long long bal = __ballot(kp_score(p) > 0);
            if (lane == 0) s_bal[c * NW + wave] = bal;
        }
    }
    __syncthreads();
    {   // exclusive prefix over the n_cells <= 1024 cells: a contiguous run of cells per thread
        const int per = (n_cells + NT - 1) / NT, c0 = tid * per, c1 = min(c0 + per, n_cells);
        int v = 0;
        for (int c = c0; c < c1; c++) v += __popcll(s_bal[c]);
        const int incl = wave_inclusive_scan_i32(v);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int t = s_wtot[w];
            if (w < wave) base += t;
            tot += t;
        }
        int run = base + incl - v;
        for (int c = c0; c < c1; c++) { s_base[c] = run; run += __popcll(s_bal[c]); }
        if (tid == 0) s_total = tot;
    }
    __syncthreads();
    const int total = s_total;
    if (NC > 0) {
#pragma unroll
        for (int c = 0; c < NR; c++) {
            const int j = c * NT + tid;
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
            const int j = c * NT + tid;
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
    for (int t = tid; t <= T; t += NT) tp[t] = compact_pos_of(t, T, total, s_bal, s_base);
    // first keypoint of every tile row (+ the end of each level), per-level counts
    for (int t = tid; t < g.row_tab_len; t += NT) {
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
        const int per = (EN + NT - 1) / NT, t0 = tid * per, t1 = min(t0 + per, EN);
        int sum = 0;
        for (int t = t0; t < t1; t++) sum += s_epi[t];
        const int incl = wave_inclusive_scan_i32(sum);
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < NW; w++)
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
    for (int i0 = tid; i0 < total; i0 += 4 * NT) {
        unsigned long long p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = i0 + NT * u < total ? kout[i0 + NT * u] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + NT * u >= total) break;
            const int lvl = kp_level(p[u]);
            const float sc = s_scale[lvl];
            const int yi = (int)((float)kp_y(p[u]) * sc), xi = (int)((float)kp_x(p[u]) * sc);      // the level-0 coordinates k_describe packs (K11)
            const int slot = atomicAdd(&s_epi[lvl * EH + min(yi, EH - 1)], 1);
            ee[slot] = make_int2(i0 + NT * u, (xi & 0xFFFF) | (yi << 16));
        }
    }
}


} // namespace jsorb
