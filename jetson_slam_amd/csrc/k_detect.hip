// k_detect.hip - fused FAST score + 3x3 NMS + per-tile arg-max ("feature culling"), all levels and all images
// of a batch in ONE launch.  The int32 score plane of the reference never exists in HBM: a workgroup stages an
// image tile (+4 px halo) in LDS, keeps the scores of its tile group (+1 px NMS halo) in LDS as u16 and emits
// one packed (x, y, score) candidate per tile.
//
// Semantics restated (bit-exact):
//   K2 FASTComputeScoreGPU_patternSize_16_lookup_mask  src/cuda/orb_FAST_compute_score.cu:1412-1560
//      mask test, two early rejects (ring px 4/12, then 0/8), 16-bit brighter/darker masks, bounded-arc LUT,
//      score = sum |p_k - v| ; written only for the interior [20,H-20) x [20,W-20) ; 0 elsewhere.
//   K3 Tile_unrolling_reduction_kernel_v2              src/cuda/orb_FAST_apply_NMS_G.cu:1178-1384
//      score kept iff >= its 8 neighbours; per tile the maximum wins; ties are resolved by the reference's thread
//      layout: within a column by (ty, k) = ((y - r*th) % n_ty, (y - r*th) / n_ty) lexicographic, across columns
//      by the ceil-halving tree over shared memory including its stale-slot re-reads.
// MI355X design:
//   * phase 0: the image tile is staged with 16-byte global loads / ds_write_b128 (a CU retires one vector-memory
//     wave-instruction per ~16 clk whatever its width; 4-byte loads capped the first version at ~1.5 TB/s).
//   * phase 1 (all pixels): the two early rejects, branch-free, 4 pixels per lane from 5 aligned LDS dwords.  Survivors
//     (~18 % on the benchmark images) are compacted into per-wave LDS work lists with wave64 __ballot + popcount
//     prefixes (no atomics), so phase 2 runs on dense waves.
//   * phase 2 (survivors, each wave on its own list, no barrier in between): 16-pixel ring, LUT bit test (8 KB bit table,
//     L1/L2 resident), SAD score (v_sad_u8) -> LDS; pixels with a positive score are re-compacted in place.
//   * phase 3 (positive scores only): 3x3 NMS from LDS and one ds_max_u32 per column with the key
//     (score << 16 | 0xFFFF - rank), rank = ty * 256 + k : the max key IS the reference's column winner.
//   * phase 4: the reference's horizontal tree replayed literally on <= 128 column slots in LDS.
// Round 5 - the COMPACT form (template parameter CP, what batch handles run): the kernel's occupancy is decided by LDS (30 VGPRs, but 33 KB per
// 4-wave workgroup = 4 waves per SIMD, half of a wave's life spent waiting), and 260 of the ~420 LDS bytes a band row costs were the u16 score
// plane, >= 93 % zeros, alive from phase 0 on.  In the compact form a positive's score travels in its list entry (score << 16 | row << 8 | column, a
// workgroup-wide u32 pool filled with one LDS atomic per ring-test chunk), and the plane is built only when phases 1 + 2 are over - ON TOP of the
// image tile and the survivor lists, which are dead by then: the workgroup zeroes the plane, scatters the pool into it, and the NMS reads the plane
// as before (its work spread over all 256 threads by pool index).  21 KB instead of 33 KB at the SAME band heights = 7 workgroups per CU
// (timing-only knock-out first: -27 % kernel time, profiles/r05_detect_occupancy_knockout.txt).  The benchmark images have 3-7 % positives on average
// and up to 18 % in single bands (checker patches on the coarse levels); the pool holds >= 10 % of a band's pixels, and what does not fit SPILLS into a
// chunk of global memory the workgroup borrows from the handle's arena: entry i of a band lives in the pool for i < pos_cap, else at chunk[i - pos_cap].
// The arena has a chunk for every workgroup that can be resident (per XCD: claimed with a compare-and-swap on a flag, returned at the end), and a
// chunk holds a whole band's pixels - so the compact form needs no second pass and no fallback: an image of pure noise costs L2 round trips, nothing else.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "jsorb_launch.h"
#include "k_blur_body.h"

namespace jsorb {

// LDS image row stride (bytes).  A tile group is at most 128 px wide; + 4 px halo each side + <= 15 alignment bytes <= 151.
// A compile-time stride turns the 16 ring offsets of phase 2 into immediate LDS offsets.
#define DET_S 160
// survivor-list capacity of one wave (entries) and the most entries one early-reject step appends (4 pixel slots x 64 lanes)
#ifndef DET_LIST_CAP
#define DET_LIST_CAP 640
#endif
#define DET_LIST_STEP 256
// waves per workgroup: they share the staged tile and take its region rows in turn
#ifndef DET_NW
#define DET_NW 4
#endif
#define DET_THREADS (64 * DET_NW)
// compact form: survivor-list capacity of one wave (entries) and the capacity of its list of positives (u32 entries: score << 16 | ry << 8 | rx)
#ifndef DET_CP_LIST_CAP
#define DET_CP_LIST_CAP 384
#endif
// The positives of a workgroup share ONE pool in LDS: at least DET_POS_PERMILLE of the band's region pixels (a band is only as tall as that allows,
// fill_detect_layout) and whatever else the LDS budget leaves; a test build caps it with DET_POS_MAX.  Positives on the benchmark images:
// 3-7 % of the pixels on average, up to 18 % in single bands of the coarse levels (checker patches).  What does not fit spills into a chunk of
// global memory (below), so a small pool costs a few L2 round trips in the dense bands, not a second pass.
#define DET_POS_MIN 256
#ifndef DET_POS_MAX
#define DET_POS_MAX 8192
#endif
#ifndef DET_POS_PERMILLE
#define DET_POS_PERMILLE 100
#endif
// LDS budget of a compact workgroup in granules of 1280 B: 17 or 18 = 7 workgroups per CU (28 waves).  Round 6 (with the 256-thread k_compact and the
// alternating lane order): 17 granules 126.1 k against 18 granules 125.1 k pairs/s at C2, three alternating runs on one box (C5: 36.7 k both) - the
// bands are the same (the pools shrink by 320 entries), but seven workgroups now leave 11.5 KB of a CU's LDS free: room for one k_blur workgroup (10 KB)
// of another lane next to them; 16 granules (8 workgroups per CU) 125.4 k.  Round 5 (C2, pairs/s of the 4-lane pipeline /
// k_detect ms per step, full-plane form 121.3 k / 0.538 on that box; tools/micro/r5_exp9.sh): 17 granules with a pool of >= 18 % of the band's
// pixels 119.0 k / 0.419, 18 / 18 % 119.4 k / 0.416, 17 / 10 % (level 1 gets a third tile row per band: 202 instead of 220 workgroups per image)
// 118.4 k / 0.419, 18 / 10 % 120.5 k / 0.411, 19 / 10 % (6 workgroups per CU) 119.4 k / 0.440, 17 / 6 % 119.0 k / 0.415.
#ifndef DET_CP_GRANULES
#define DET_CP_GRANULES 17
#endif
#define DET_CP_BUDGET (DET_CP_GRANULES * 1280)
// The spill arena of a handle: 8 XCDs x DET_ARENA_SLOTS chunks.  A workgroup only ever touches the chunks of the XCD it runs on (s_getreg XCC_ID), so a
// chunk's data passes through ONE L2 - the L2s of different XCDs are not coherent with each other inside a kernel.  At most 32 CUs x 8 workgroups of
// this kernel are resident per XCD (4 waves each, 32 wave slots per CU): 320 slots can never all be taken.
#define DET_ARENA_XCDS 8
#ifndef DET_ARENA_SLOTS
#define DET_ARENA_SLOTS 320
#endif

struct DetectLds {
    int img_stride;      // bytes per LDS image row (multiple of 16)
    int img_rows;        // th + 8
    int score_w;         // k*tw + 2
    int score_stride;    // u16 elements per row of the score plane (compact form: rounded up to an even number - rows start on dwords)
    int score_rows;      // th + 2
    int list_cap;        // survivor-list capacity of ONE wave (entries)
    int pos_cap;         // compact form: entries of the workgroup's pool of positives
    size_t off_score, off_list, off_pos, off_colkey, off_tree, total;
};

// compact = 0: image tile | score plane | survivor lists | keys | tree        (every region alive from phase 0 to phase 3)
// compact = 1: [image tile | survivor lists] overlaid by the score plane | lists of positives | keys | tree | overflow flag
__host__ __device__ inline DetectLds detect_lds_layout(int th, int tw, int k_tiles, int ranked, int compact, size_t budget = DET_CP_BUDGET)
{
    DetectLds d;
    const int ktw = k_tiles * tw;
    d.img_stride = DET_S;                          // fixed: window is 16-byte aligned on the left (<= 15 extra bytes), ktw <= 128
    d.img_rows = th + 8;
    d.score_w = ktw + 2;
    d.score_stride = compact ? (d.score_w + 1) & ~1 : d.score_w;
    d.score_rows = th + 2;
    // a wave owns <= 2*ceil(rows/8) rows of the score region.  The list is CAPPED (the worst case - every pixel survives the early
    // rejects - would cost 7.8 KB at tile 30 and hold the kernel at 7 workgroups per CU): when a wave's list could not take another
    // early-reject step (DET_LIST_STEP entries), the wave runs its ring test on what it has - only positives stay in the list - and
    // goes on; if even the positives do not fit, the wave falls back to a dense scan of its rows in phase 3 (full-plane form; in the compact
    // form positives never stay in this list).
    const int full = 2 * ((d.score_rows + 2 * DET_NW - 1) / (2 * DET_NW)) * d.score_w;
    const int cap = compact ? DET_CP_LIST_CAP : DET_LIST_CAP;
    d.list_cap = full <= cap ? full : cap;
    size_t o = (size_t)d.img_stride * d.img_rows;
    o = (o + 15) & ~(size_t)15;
    if (compact) {
        d.off_score = 0;
        d.off_list = o;
        o += (size_t)(d.list_cap + 64) * DET_NW * 2;      // + 64 dump slots per wave: lanes without a survivor store there (no exec masking around the list appends)
        const size_t plane = (size_t)d.score_stride * ((d.score_rows + 1) & ~1) * 2;      // (an even number of rows: a wave zeroes its rows in pairs)
        if (o < plane) o = plane;
        o = (o + 15) & ~(size_t)15;
        d.off_pos = o;
        // the pool of positives: DET_POS_PERMILLE of the region's pixels, or more if the budget has room left behind the plane and the fixed part
        const size_t fixed = 128 * 4 + (ranked ? 256 : 128 * 8) + 16;
        long cap = ((long)budget - (long)o - (long)fixed) / 4;
        const long want = ((long)d.score_w * d.score_rows * DET_POS_PERMILLE + 999) / 1000;
        cap = cap < want ? want : cap;
        cap = cap < DET_POS_MIN ? DET_POS_MIN : cap;
        cap = cap > DET_POS_MAX ? DET_POS_MAX : cap;
        d.pos_cap = (int)cap;
        o += (size_t)d.pos_cap * 4;
    } else {
        d.pos_cap = 0;
        d.off_score = o;
        o += (size_t)d.score_w * d.score_rows * 2;
        o = (o + 15) & ~(size_t)15;
        d.off_list = o;
        o += (size_t)(d.list_cap + 64) * DET_NW * 2;
        o = (o + 15) & ~(size_t)15;
        d.off_pos = o;
    }
    d.off_colkey = o;
    o += 128 * 4;
    d.off_tree = o;
    o += ranked ? 256 : 128 * 8;                   // column priorities (rank[128], inv[128]) or the literal tree's 128 slots
    if (compact) o += 16;                          // the workgroup's overflow flag and the fill count of its pool
    d.total = o;
    return d;
}
__host__ __device__ inline int detect_flush_at(const DetectLds &d)
{
    // a list that holds the worst case (every pixel of the wave's rows survives) never needs the early ring test
    const int full = 2 * ((d.score_rows + 2 * DET_NW - 1) / (2 * DET_NW)) * d.score_w;
    return d.list_cap >= full ? 0x7fffffff : d.list_cap - DET_LIST_STEP;
}

// Tile rows per workgroup.  The launch-wide LDS size is set by the level with the largest tiles (level 0 unless the tile size is
// fixed); the levels above it have smaller tiles and their workgroups would leave most of that allocation unused while paying the
// same fixed cost (scalar prologue, staging passes, barriers, list bookkeeping - about 150 VALU + 500 SALU per wave, a third of a
// wave's instructions at 8-row tiles).  Those levels put as many tile rows into one workgroup as fit into the SAME allocation.
#ifndef DET_MAX_R
#define DET_MAX_R 8
#endif
void fill_detect_layout(Geometry &g)
{
    const int cp = g.det_compact;
    size_t budget = 0;
    for (int i = 0; i < g.L; i++) budget = std::max(budget, detect_lds_layout(g.lv[i].th, g.lv[i].tw, g.lv[i].k_tiles, g.lv[i].tree_rank_ok, cp, 0).total);      // (budget 0: the smallest pool the level may have)
    // Full-plane form: the allocation may grow to just under a fifth of a CU's LDS if that lets more tile rows share a workgroup: every wave pays a
    // prologue of ~230 scalar + ~150 vector instructions, and the scalar pipe is nearly as busy as the vector pipes in this kernel.  Measured at the
    // end of round 3 (pairs/s at C2 / C3 / C5): 7 workgroups per CU (the round-2 choice) 107.1 / 80.6 / 30.8 k, 6: 109.7 / 84.9 / 31.7 k,
    // 5: 111.7 / 85.5 / 32.0-32.3 k, 4: 109.5 / 85.7 / 31.6 k.
    // Compact form (round 5): the same band heights cost ~22 KB with a pool for >= 10 % positives: the budget is 18 granules - 7 workgroups = 7 waves per SIMD.
    if (const char *b7 = experiment_env("JSORB_DETECT_BUDGET")) budget = std::max(budget, (size_t)atoi(b7));
    else budget = std::max(budget, cp ? (size_t)DET_CP_BUDGET : (size_t)(160 * 1024 / 5 - 256));
    int dblk = 0;
    for (int i = 0; i < g.L; i++) {
        LevelDesc &lv = g.lv[i];
        int R = 1;
        // only the arg-max form of the tile reduction (tree_rank_ok) handles several tile rows; keys: R * k_tiles <= 128 slots, row index < 256
        while (lv.tree_rank_ok && R < DET_MAX_R && R < lv.nth && (R + 1) * lv.k_tiles <= 128 && (R + 1) * lv.th + 2 <= 255 &&
               detect_lds_layout((R + 1) * lv.th, lv.tw, lv.k_tiles, 1, cp, 0).total <= budget && !experiment_env("JSORB_DETECT_NO_BANDS") && !g.latency)      // single-image handles: one tile row per workgroup
            R++;
        lv.det_R = R;
        lv.detect_blk0 = dblk;
        dblk += ((lv.nth + R - 1) / R) * lv.groups_per_row;
        const DetectLds d = detect_lds_layout(R * lv.th, lv.tw, lv.k_tiles, lv.tree_rank_ok, cp, budget);
        lv.det_score_w = d.score_w; lv.det_score_stride = d.score_stride; lv.det_score_rows = d.score_rows; lv.det_img_rows = d.img_rows; lv.det_list_cap = d.list_cap;
        lv.det_flush_at = detect_flush_at(d);
        lv.det_pos_cap = d.pos_cap;
        lv.det_off_score = (int)d.off_score; lv.det_off_list = (int)d.off_list; lv.det_off_pos = (int)d.off_pos; lv.det_off_colkey = (int)d.off_colkey; lv.det_off_tree = (int)d.off_tree;
    }
    g.detect_blocks = dblk;
}

// dynamic LDS of the launch (the handle's form)
static size_t detect_lds_bytes_form(const Geometry &g, int compact)
{
    size_t m = 0;
    for (int i = 0; i < g.L; i++) {
        DetectLds d = detect_lds_layout(g.lv[i].det_R * g.lv[i].th, g.lv[i].tw, g.lv[i].k_tiles, g.lv[i].tree_rank_ok, compact, 0);
        if (compact) d.total += (size_t)(g.lv[i].det_pos_cap - d.pos_cap) * 4;      // the pool as fill_detect_layout sized it
        if (d.total > m) m = d.total;
    }
    return m;
}
size_t detect_lds_bytes(const Geometry &g) { return detect_lds_bytes_form(g, g.det_compact); }
// entries of one spill chunk: the largest band region of the handle (every pixel a positive)
int detect_spill_chunk_entries(const Geometry &g)
{
    int m = 1;
    for (int i = 0; i < g.L; i++) m = std::max(m, g.lv[i].det_score_w * g.lv[i].det_score_rows);
    return (m + 63) & ~63;
}
size_t detect_arena_bytes(const Geometry &g) { return (size_t)DET_ARENA_XCDS * DET_ARENA_SLOTS * detect_spill_chunk_entries(g) * sizeof(unsigned); }
size_t detect_arena_flag_words() { return (size_t)DET_ARENA_XCDS * DET_ARENA_SLOTS; }
// workgroups of the compact k_detect that may be resident at once on this many CUs (4 waves each, 32 wave slots per CU) must not exceed the arena's chunks
// (an XCD of a gfx942 / gfx950 part has at most 40 CUs, 32-38 of them active, whatever the partition mode: 40 x 8 resident workgroups = the 320 slots of
// an XCD's partition; a device that reports more CUs than 8 such XCDs hold is not one this layout was made for)
bool detect_arena_covers(int compute_units) { return compute_units >= 1 && compute_units <= DET_ARENA_XCDS * (DET_ARENA_SLOTS / 8); }
int detect_pos_cap(const Geometry &g, int level) { return g.lv[level].det_pos_cap; }

// Early rejects on 6-bit pixels (k_detect phase 1, SWAR form): with q(x) = x >> 2 and t4 = (th + 1) >> 2,
//   p > v + th  =>  q(p) - q(v) >= t4        and        p < v - th  =>  q(v) - q(p) >= t4
// (floor((v + th + 1) / 4) >= floor(v / 4) + floor((th + 1) / 4)), so the 6-bit tests accept a SUPERSET of what the exact tests accept, and the
// compass / near-pair combinations of phase 1 are monotone in their flags.  In 8-bit fields the sums q(p) + (128 - t4 - q(v)) and
// (128 - t4 + q(v)) - q(p) stay inside [1, 191]: no carry or borrow between the four pixels of a dword, bit 7 of a field IS the flag.
// Returns t4 if the property holds for every (v, p) - checked exhaustively, not assumed - and the test still rejects something (t4 >= 2), else 0.
int detect_swar6_threshold(int threshold)
{
    const int thc = std::min(threshold, 256), t4 = (thc + 1) >> 2;
    if (t4 < 2 || experiment_env("JSORB_DETECT_EXACT_REJECT")) return 0;
    const int cb = 128 - t4;
    for (int v = 0; v < 256; v++)
        for (int p = 0; p < 256; p++) {
            const int y = (p >> 2) + (cb - (v >> 2)), z = (cb + (v >> 2)) - (p >> 2);
            if (y < 0 || y > 255 || z < 0 || z > 255 || cb - (v >> 2) < 0 || cb + (v >> 2) > 255) return 0;      // a field would carry into its neighbour
            if (p > v + threshold && !(y & 0x80)) return 0;
            if (p < v - threshold && !(z & 0x80)) return 0;
        }
    return t4;
}

// Bit order of the ring masks inside k_detect's phase 2 (see there): the 32-bit word has ring pixel 4j + b at bit 8b + 7 - j; squeezed
// to 16 bits for the arc LUT, pixel 4j + b sits at bit 4b + 3 - j.  build_lut_bits() on the host stores the LUT in this order.
__host__ __device__ __forceinline__ int ring_bit_of_pixel(int k) { return 4 * (k & 3) + 3 - (k >> 2); }
__device__ __forceinline__ unsigned ring_word_to_index(unsigned w)
{
    const unsigned x = w >> 4;                 // nibbles at bits 0-3, 8-11, 16-19, 24-27
    const unsigned y = x | (x >> 4);           // byte 0 = nibble0 | nibble1 << 4, byte 2 = nibble2 | nibble3 << 4
    return (y & 0xFFu) | ((y >> 8) & 0xFF00u);
}
int detect_ring_bit_of_pixel(int k) { return ring_bit_of_pixel(k); }


// one workgroup of k_detect: image b, workgroup blk of the image's g.detect_blocks (a kernel of its own for batches, one half of the fused
// k_detect_blur launch for single frames - both below)
// A workgroup's first spill: claim a free chunk of the XCD the wave runs on (compare-and-swap on the chunk's busy flag, probing from a slot that depends
// on the workgroup), publish it in the workgroup's LDS word; if another wave of the workgroup was faster, give the chunk back and take that one.
// Returns chunk index + 1.  Deliberately NOT inlined: the probing loop sits in the middle of the ring pass, where the scalar registers are scarce -
// inlined, the kernel went from 82 to 106 SGPRs and its hot loops reloaded kernel arguments (k_detect 215 -> 247 us per 128 images).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "k_detect's spill arena reads HW_REG_XCC_ID (hwreg 20): gfx942 / gfx950 only"
#endif
__device__ __attribute__((noinline)) unsigned detect_claim_chunk(unsigned *spill_flags, unsigned *s_chunk, unsigned blk, unsigned b)
{
    unsigned ch = __hip_atomic_load(s_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ch != 0u) return ch;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | ((4 - 1) << 11)) & (DET_ARENA_XCDS - 1);      // HW_REG_XCC_ID[3:0]
    unsigned *const fl = spill_flags + xcc * DET_ARENA_SLOTS;
    unsigned slot = (blk * 2654435761u + b * 40503u) % DET_ARENA_SLOTS;
    int tries = 0, sweeps = 0;
#pragma clang loop unroll(disable)
    while (atomicCAS(fl + slot, 0u, 1u) != 0u) {
        slot = slot + 1 == DET_ARENA_SLOTS ? 0 : slot + 1;
        if (++tries == DET_ARENA_SLOTS) {
            // A whole sweep without a free chunk: more holders than the partition has slots.  That takes workgroups parked with their chunks (the driver
            // context-saved their waves under queue oversubscription) or a test build with a handful of slots: back off and keep probing - every holder gives
            // its chunk back at the end of its NMS.  Only a partition that stays full for about a second (flags corrupted) ends the kernel loudly.
            tries = 0;
            __builtin_amdgcn_s_sleep(127);
            if (++sweeps > 4096) __builtin_trap();
        }
    }
    const unsigned mine = xcc * DET_ARENA_SLOTS + slot + 1u;
    const unsigned old = atomicCAS(s_chunk, 0u, mine);
    if (old == 0u) return mine;
    __hip_atomic_store(fl + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (nothing was written to the chunk)
    return old;
}

// CP: the compact form (see the head of the file); spill / spill_flags / chunk_entries: the handle's arena of spill chunks (CP only)
template <bool HAS_MASK, bool COMPASS, bool SWAR, bool CP>
__device__ __forceinline__ void detect_workgroup(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *mask_slab,
                                                 const uint32_t *__restrict__ lut_bits, unsigned long long *tile_out, int b, int blk,
                                                 unsigned *spill = nullptr, unsigned *spill_flags = nullptr, int chunk_entries = 0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i32(tid >> 6);      // (tid >> 6 is wave-uniform, but only a readfirstlane proves it to the compiler: loop counters and list sizes derived from it then live in SGPRs)
    // workgroup descriptor (level, tile row, tile group) from the host-built table behind the LUT: one scalar load instead of a
    // chain of dependent ones - the kernel is sensitive to the latency of this prologue (no vector work can start before it)
    const unsigned wd = ctab_load(lut_bits, CTAB_DETECT + blk);
    const int lvl = (int)(wd & 15u), r = (int)((wd >> 4) & 0x3FFFu), grp = (int)(wd >> 18);
    const LevelDesc &lv = g.lv[lvl];
    const int H = lv.H, W = lv.W, th1 = lv.th, tw = lv.tw, R = lv.det_R;
    asm volatile("" ::"s"(lv.img_off), "s"(lv.pitch), "s"(lv.k_tiles), "s"(lv.det_img_rows), "s"(H), "s"(th1), "s"(R));       // one round of loads
    const int ktw = lv.k_tiles * tw;
    const int xg0 = grp * ktw;            // first image column of the tile group
    const int tr0 = r * R;                // first tile row of the band
    const int th = R * th1;               // rows of the band: R tile rows of th1 rows each (R = 1 on the levels with the largest tiles)
    const int y0 = tr0 * th1;             // first image row of the band
    DetectLds L;
    int flush_at;                                         // wave-uniform; INT_MAX when the list holds the worst case (no early ring test)
    L.img_stride = DET_S; L.img_rows = lv.det_img_rows; L.score_w = lv.det_score_w; L.score_stride = lv.det_score_stride; L.score_rows = lv.det_score_rows; L.list_cap = lv.det_list_cap;
    L.off_score = (size_t)lv.det_off_score; L.off_list = (size_t)lv.det_off_list; L.off_pos = (size_t)lv.det_off_pos; L.off_colkey = (size_t)lv.det_off_colkey; L.off_tree = (size_t)lv.det_off_tree;
    flush_at = lv.det_flush_at;
    unsigned char *s_img = smem;
    unsigned short *s_score = reinterpret_cast<unsigned short *>(smem + L.off_score);
    unsigned short *s_list = reinterpret_cast<unsigned short *>(smem + L.off_list);
    unsigned *s_pos = reinterpret_cast<unsigned *>(smem + L.off_pos);                              // CP: the workgroup's pool of positives
    unsigned *s_overflow = reinterpret_cast<unsigned *>(smem + L.off_tree + (lv.tree_rank_ok ? 256 : 1024));      // CP: [1] positives of the band; [2] the band's spill chunk + 1 (0: none)
    const int pos_cap = CP ? lv.det_pos_cap : 0;
    unsigned *s_colkey = reinterpret_cast<unsigned *>(smem + L.off_colkey);
    unsigned long long *s_tree = reinterpret_cast<unsigned long long *>(smem + L.off_tree);

    // A band / tile group that lies entirely in the image's 20-pixel border has no pixel to test: the sliver tiles at the right and bottom edge of a level
    // (ceil(W / tw) tiles: at the EuRoC geometry 32 of an image's 223 workgroups - e.g. level 1, 627 = 5 x 125 + 2 columns, has a sixth tile group of two
    // columns in each of its eight bands).  Such a workgroup writes its tiles' empty records (what phase 4 writes for a tile without a positive) and is done -
    // no staging, no barriers, no plane: k_detect -2 %, C2 +0.5 %, tile 58 +1.2 %, C3 +0.7 % (round 6).  (Wave-uniform scalars; arg-max form only: the literal tree's levels take the ordinary path.)
    {
        const int xs_e = (xg0 - 4) & ~15, c0_e = (xg0 - 1) - xs_e;
        const int c_lo_e = max(c0_e, JSORB_BORDER - xs_e), c_hi_e = min(c0_e + L.score_w, W - JSORB_BORDER - xs_e);
        const int ry_lo_e = max(0, JSORB_BORDER - (y0 - 1)), ry_hi_e = min(L.score_rows - 1, H - JSORB_BORDER - 1 - (y0 - 1));
        if (lv.tree_rank_ok && (c_hi_e <= c_lo_e || ry_hi_e < ry_lo_e)) {
            const int kt = lv.k_tiles;
            int trow = 0;
            for (int rr = 1; rr < R; rr++) trow += tid >= rr * kt;      // (wave-uniform trip count <= DET_MAX_R - 1)
            const int tcol = tid - trow * kt;
            const int tr = tr0 + trow;
            if (tid < R * kt && tr < lv.nth && xg0 + tcol * tw < W)
                tile_out[(size_t)b * g.T + lv.tile_off + tr * lv.ntw + grp * kt + tcol] =
                    (unsigned long long)(((unsigned)((y0 + trow * th1) & 0xFFFF) << 16) | (unsigned)((xg0 + tcol * tw) & 0xFFFF));
            return;
        }
    }
    int pitch;
    const uint8_t *img = level_ptr_uniform(g, src, slab, b, lvl, lv.pitch, lv.img_off, pitch);

    // ---- phase 0: stage image rows [y0-4, y0+th+4) x cols [xs, xs+S) with 16-byte loads ; zero the score tile ----
    // (4-byte loads cap a CU at ~1/4 of its HBM rate: the vector-memory pipeline retires one wave-instruction per
    //  ~16 clk whatever its width, so staging uses global_load_dwordx4 / ds_write_b128 throughout)
    constexpr int S = DET_S;
    const int xs = (xg0 - 4) & ~15;                       // 16-byte aligned (may be negative)
    constexpr int nq16 = S >> 4;                          // 10 units of 16 B per LDS row
    constexpr int rpp = DET_THREADS / nq16;               // 25 rows per pass (250 of the 256 threads)
    {
        // a thread keeps its column and walks down the tile with a constant pointer stride: no per-item index arithmetic
        const int dx = tid % nq16, ly0 = tid / nq16;
        const int x = xs + 16 * dx;
        const bool x_ok = tid < rpp * nq16 && x >= 0 && x + 16 <= pitch;
        const uint8_t *p16 = img + (ptrdiff_t)(y0 - 4 + ly0) * pitch + x;
        uint4 *dst = reinterpret_cast<uint4 *>(s_img) + tid;
        int y = y0 - 4 + ly0;
        // ALL of a thread's rows are requested before the first one is stored (round 6): written as one loop - load, store, next row - the compiler waits
        // for every load before the next is issued, three to five dependent memory round trips in front of the workgroup's first barrier.  A band has at
        // most DET_STAGE_MAX * 25 rows (tiles of up to 128 rows + 8: 6 passes); taller ones take the rest in the plain loop.
        constexpr int DET_STAGE_MAX = 6;
        uint4 sv[DET_STAGE_MAX];
#pragma unroll
        for (int k = 0; k < DET_STAGE_MAX; k++) {
            sv[k] = make_uint4(0, 0, 0, 0);
            const int yk = y + k * rpp;
            if (ly0 + k * rpp < L.img_rows && x_ok && yk >= 0 && yk < H) sv[k] = *reinterpret_cast<const uint4 *>(p16 + (size_t)k * rpp * pitch);
        }
#pragma unroll
        for (int k = 0; k < DET_STAGE_MAX; k++)
            if (ly0 + k * rpp < L.img_rows && tid < rpp * nq16) dst[k * rpp * nq16] = sv[k];
        p16 += (size_t)DET_STAGE_MAX * rpp * pitch;
        dst += DET_STAGE_MAX * rpp * nq16;
        y += DET_STAGE_MAX * rpp;
        for (int ly = ly0 + DET_STAGE_MAX * rpp; ly < L.img_rows; ly += rpp) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x_ok && y >= 0 && y < H) v = *reinterpret_cast<const uint4 *>(p16);
            if (tid < rpp * nq16) *dst = v;
            p16 += (size_t)rpp * pitch;
            dst += rpp * nq16;
            y += rpp;
        }
    }
    {
        if constexpr (!CP) {                              // (compact form: the plane does not exist yet - it is built after phase 2, where the image tile is now)
            const int nsc = (L.score_w * L.score_rows * 2 + 15) >> 4;
            uint4 *z = reinterpret_cast<uint4 *>(s_score);
            for (int i = tid; i < nsc; i += DET_THREADS) z[i] = make_uint4(0, 0, 0, 0);
        } else if (tid < 3) s_overflow[tid] = 0u;
        if (tid < 128) s_colkey[tid] = 0;
        if (lv.tree_rank_ok && tid < 64) reinterpret_cast<unsigned *>(s_tree)[tid] = lut_bits[ctab_tree(g) + 64 * lvl + tid];      // column priorities
    }
    __syncthreads();

    // ---- phase 1: the two early rejects on every pixel of the (th+2) x (ktw+2) score region, 4 pixels per lane ----
    // LDS column c <-> image x = xs + c ; region column rx <-> c = c0 + rx.  A lane owns one aligned LDS dword (4 pixels)
    // and reads 5 dwords (left/centre/right, row-3, row+3); a wave covers two region rows per step when a row fits in 32
    // dwords.  Survivors go to a per-wave list (no atomics): 4 ballots + popcount prefix per step.
    const int threshold = g.threshold;
    const int c0 = (xg0 - 1) - xs;                        // LDS column of region column 0   (3 <= c0 <= 18)
    const int q0 = c0 >> 2;
    const int nq = ((c0 + L.score_w - 1) >> 2) - q0 + 1;  // dwords per region row
    const int c_lo = max(c0, JSORB_BORDER - xs), c_hi = min(c0 + L.score_w, W - JSORB_BORDER - xs);
    const unsigned c_span = c_hi > c_lo ? (unsigned)(c_hi - c_lo) : 0u;
    const uint8_t *mask = HAS_MASK ? mask_slab + lv.img_off : nullptr;
    const int two_rows = nq <= 32 ? 1 : 0;
    const int sub = two_rows ? (lane >> 5) : 0;
    const int q = two_rows ? (lane & 31) : lane;
    const int rows_per_step = two_rows ? 2 : 1;
    unsigned short *my_list = s_list + wave * (L.list_cap + 64);
    unsigned short *const dump_ptr = my_list + L.list_cap + lane;      // where a lane without a survivor stores (never read)
    int n_mine = 0;                                       // wave-uniform
    // everything that depends only on the lane's column is loop-invariant: the dword it owns and the nibble of its pixels
    // that lie inside [c_lo, c_hi) (interior columns of this tile group)
    const int qq = q < nq ? q0 + q : q0;                  // keep LDS addresses in range for idle lanes
    const int cb = 4 * qq;
    // cm[h]: sign-bit positions (bit 15 = pixel 2h, bit 31 = pixel 2h+1) of the lane's pixels that lie inside [c_lo, c_hi);
    // cmF: the same set as bit 7 of byte t (the flag word of the SWAR form)
    unsigned cm[2] = {0u, 0u}, cmF = 0u;
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (q < nq && (unsigned)(cb + t - c_lo) < c_span) { cm[t >> 1] |= (t & 1) ? 0x80000000u : 0x8000u; cmF |= 0x80u << (8 * t); }
    // two pixels per instruction in the 16-bit halves of a dword (v_pk_*_i16); every test ends as the sign bit of a packed
    // difference.  th >= 256 behaves like 256 (every pixel is "near").
    typedef short s2 __attribute__((ext_vector_type(2)));
    const int thc = min(threshold, 256);
    const s2 th_pk = (s2){(short)thc, (short)thc};
#define PK(hi, lo, sel) __builtin_bit_cast(s2, __builtin_amdgcn_perm((hi), (lo), (sel)))
    // Phases 1 and 2 share one loop: early-reject steps append to the wave's list; when the list could not take another step (or the
    // wave has done all its rows) the ring test runs on the pending entries [n_pos, n_mine) and leaves only positives [0, n_pos).
    const int lx_off = c0;
    const unsigned char *const lane_img = s_img + sub * S + 4 * qq;                      // the lane's dword in image row `sub` of the LDS tile
    const int ry_lo = max(0, JSORB_BORDER - (y0 - 1)), ry_hi = min(L.score_rows - 1, H - JSORB_BORDER - 1 - (y0 - 1));      // region rows inside the image's interior (wave-uniform)
    const int e_lane = (sub << 8) + (cb - c0);                                           // list entry of the lane's first pixel in row `sub`
    const int min_pop = g.lut_min_pop;
    int n_pos = 0;                                        // wave-uniform: positives at the front of the list
    bool dense = false;                                   // wave-uniform, full-plane form: the positives overflowed the list, phase 3 scans this wave's rows densely
    // ---- phase 2 (called when the list could not take another early-reject step, and once at the end): full 16-ring test + score,
    // each wave on ITS OWN survivor list (no barrier after phase 1) ----
    // Survivors whose arc test succeeds are compacted in place (ballot + popcount) to the front of the same list: writes of a
    // step land at or below the indices the step has just read, and LDS operations of one wave execute in order.
    // Compact form: the pending survivors are the whole list [0, n_mine); a pass takes full chunks of 64 from its END (the order of the entries is
    // free) and leaves the < 64 others pending unless it is the wave's last pass, so every ring test but the last one runs on a full wave; a positive goes
    // to the workgroup's pool of positives (or its spill chunk) with its score.
    auto ring_pass = [&](bool last) {
        (void)last;
        for (int i0 = CP ? n_mine - 64 : n_pos; CP ? (n_mine > 0 && (last || i0 >= 0)) : i0 < n_mine; i0 += CP ? -64 : 64) {
            const int i = i0 + lane;
            bool hit = false;
            int e = 0;
            unsigned sad = 0;
            if (CP ? i >= 0 : i < n_mine) {
                e = my_list[i];
                const int ry = e >> 8, rx = e & 255;
                const unsigned char *c = s_img + (ry + 3) * S + lx_off + rx;
                const int v = c[0], vt = v + threshold, v_t = v - threshold;
                // the 16 ring pixels as 8 packed pairs (pixel 2i in the low, 2i+1 in the high half of P[i]): every test below works
                // on two pixels per instruction (v_pk_sub_i16 / v_perm / v_sad_u16)
                // (Round 6 measured the chain list entry -> address -> 17 byte reads requested ONE CHUNK AHEAD, raw bytes kept in 17 registers across the
                // iteration: k_detect 0.41 -> 0.45 ms per step, the pipeline -1.6 % - dropped, profiles/r06_experiments.txt.)
                // (ds_read_u8_d16 / _d16_hi would land the bytes in the halves of a pair register without the v_lshl_or_b32 per pair, but gfx950 runs with SRAM ECC,
                // where a d16 load rewrites the whole register - the compiler does not use them for that reason, and neither may inline assembly)
                unsigned P[8];
                P[0] = (unsigned)c[3 * S] | ((unsigned)c[3 * S + 1] << 16);
                P[1] = (unsigned)c[2 * S + 2] | ((unsigned)c[S + 3] << 16);
                P[2] = (unsigned)c[3] | ((unsigned)c[-S + 3] << 16);
                P[3] = (unsigned)c[-2 * S + 2] | ((unsigned)c[-3 * S + 1] << 16);
                P[4] = (unsigned)c[-3 * S] | ((unsigned)c[-3 * S - 1] << 16);
                P[5] = (unsigned)c[-2 * S - 2] | ((unsigned)c[-S - 3] << 16);
                P[6] = (unsigned)c[-3] | ((unsigned)c[S - 3] << 16);
                P[7] = (unsigned)c[2 * S - 2] | ((unsigned)c[3 * S - 1] << 16);
                // brighter / darker: sign bit of (vt - p) [p > vt] and of (p - v_t) [p < v_t] per 16-bit half; v_perm collects the sign
                // bytes of four pixels into one dword E_j, and the four E_j are merged with staggered shifts: pixel 4j + b ends up at
                // bit 8b + 7 - j.  The population count is taken from this word directly; the arc LUT is stored in the matching
                // bit order (build_lut_bits, ring_bit_of_pixel), so the rare lookup only squeezes the word to 16 bits.
                typedef short s2 __attribute__((ext_vector_type(2)));
                const s2 vt2 = __builtin_bit_cast(s2, (unsigned)vt * 0x10001u), v_t2 = __builtin_bit_cast(s2, ((unsigned)v_t & 0xFFFFu) * 0x10001u);
                unsigned bright = 0, dark = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const s2 p0 = __builtin_bit_cast(s2, P[2 * j]), p1 = __builtin_bit_cast(s2, P[2 * j + 1]);
                    const unsigned eb = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, (s2)(vt2 - p1)), __builtin_bit_cast(unsigned, (s2)(vt2 - p0)), 0x07050301u);
                    const unsigned ed = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, (s2)(p1 - v_t2)), __builtin_bit_cast(unsigned, (s2)(p0 - v_t2)), 0x07050301u);
                    bright |= (eb >> j) & (0x80808080u >> j);
                    dark |= (ed >> j) & (0x80808080u >> j);
                }
                // arc LUT (8 KB bit table, global): a divergent dword gather costs the vector-memory pipeline ~1 lane/clk, so masks
                // with fewer set bits than any accepted mask skip it (bright and dark are disjoint: with N_MIN >= 9 at most one
                // of them is ever looked up, usually none)
                unsigned lb = 0;
                if (min_pop >= 9) {
                    // bright and dark are disjoint subsets of 16 ring pixels: with N_MIN >= 9 at most ONE of them can reach min_pop, so one lookup per
                    // pixel decides - one load and one memory round trip per chunk of 64 ring tests, where two conditional lookups were two dependent
                    // round trips in the middle of the pass whenever a chunk held candidates of both polarities (round 6)
                    const unsigned m = __popc(bright) >= min_pop ? bright : dark;
                    if (__popc(m) >= min_pop) { const unsigned ix = ring_word_to_index(m); lb = lut_bits[ix >> 5] >> (ix & 31); }
                } else {
                    if (__popc(bright) >= min_pop) { const unsigned ix = ring_word_to_index(bright); lb = lut_bits[ix >> 5] >> (ix & 31); }
                    if (__popc(dark) >= min_pop) { const unsigned ix = ring_word_to_index(dark); lb |= lut_bits[ix >> 5] >> (ix & 31); }
                }
                hit = (lb & 1u) != 0;
                if (hit) {
                    const unsigned v2 = (unsigned)v * 0x10001u;
#pragma unroll
                    for (int k = 0; k < 8; k++) sad = __builtin_amdgcn_sad_u16(P[k], v2, sad);
                    if constexpr (!CP) s_score[__umul24(ry, L.score_w) + rx] = (unsigned short)sad;
                }
            }
            const unsigned long long bal = __ballot(hit);
            const unsigned rank_in_chunk = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            if constexpr (CP) {
                n_mine = i0 > 0 ? i0 : 0;
                if (bal != 0ull) {
                    // room in the workgroup's pool for the chunk's positives: one LDS atomic by one lane, its result through an SGPR
                    const unsigned cnt = (unsigned)__popcll(bal);
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(s_overflow + 1, cnt);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    const unsigned ent = (sad << 16) | (unsigned)e;                     // (a score is <= 16 * 255 = 4080)
                    if (__builtin_expect(base + cnt <= (unsigned)pos_cap, 1)) {
                        if (hit) (s_pos + base)[rank_in_chunk] = ent;
                    } else {
                        // The pool is full: the entries beyond it SPILL into a chunk of global memory that the workgroup borrows from the handle's arena the
                        // first time it needs one.  One lane claims a free chunk of ITS XCD (compare-and-swap on the chunk's flag, probing from a
                        // slot that depends on the workgroup); if two waves claim at the same time the loser gives its chunk back.
                        unsigned ch = 0;
                        if (lane == 0) ch = detect_claim_chunk(spill_flags, s_overflow + 2, (unsigned)blk, (unsigned)b);
                        ch = (unsigned)__builtin_amdgcn_readfirstlane((int)ch);
                        unsigned *const chunk = spill + (size_t)(ch - 1u) * (unsigned)chunk_entries;
                        const unsigned idx = base + rank_in_chunk;
                        if (hit && idx < (unsigned)pos_cap) s_pos[idx] = ent;
                        if (hit && idx >= (unsigned)pos_cap) chunk[idx - (unsigned)pos_cap] = ent;
                    }
                }
            } else {
                if (hit && !dense) (my_list + n_pos)[rank_in_chunk] = (unsigned short)e;
            }
            n_pos += __popcll(bal);
        }
        if constexpr (!CP) {
            if (dense || n_pos > flush_at) { dense = true; n_pos = 0; }          // not even the positives fit: dense scan in phase 3
            n_mine = n_pos;
        }
    };
    // The wave's early-reject steps: rows rbase (+1) of the region, every 4th step of the workgroup.  Steps whose rows all lie in the
    // image's 20-pixel border have no pixel to test (12 % of the steps at the EuRoC geometry, 30 % of the rows of the smallest level):
    // the loop runs over the others only - a plain counted loop; written as one loop with the ring test inside and `continue` for the
    // border steps, the compiler built a state machine of ~75 scalar instructions per step, as many as the vector ones.
    const int step_rows = DET_NW * rows_per_step;
    int rb_first = wave * rows_per_step;
    {
        const int rb_min = JSORB_BORDER - (y0 - 1) - (rows_per_step - 1);            // first rbase with a row at or below the border line
        static_assert(DET_NW == 4 || DET_NW == 8, "step_rows must be a power of two");
        const int sh = (DET_NW == 8 ? 3 : 2) + (two_rows ? 1 : 0);                        // log2(step_rows)
        if (rb_first < rb_min) rb_first += ((rb_min - rb_first + step_rows - 1) >> sh) << sh;
    }
    const int rb_end = min(L.score_rows, H - JSORB_BORDER - (y0 - 1));               // rbase < rb_end: the region and the image's interior
#define DET_RING_PASS(last) ring_pass(last)
    int e_cur = (rb_first << 8) + e_lane;                  // list entry of the lane's first pixel in the current step
    for (int rbase = rb_first; rbase < rb_end; rbase += step_rows) {
        if (n_mine > flush_at) DET_RING_PASS(false);
        const int ry = rbase + sub;
        const int y = y0 - 1 + ry;
        const unsigned *rowp = reinterpret_cast<const unsigned *>(lane_img + (rbase + 3) * S);
        const unsigned Dm = rowp[-1], D0 = rowp[0], Dp = rowp[1];
        const unsigned Du = rowp[-3 * (S >> 2)], Dd = rowp[3 * (S >> 2)];
        const int e0 = e_cur;                             // (rbase << 8) + e_lane: cb - c0 may be negative for the first dword; e0 + t is not
        e_cur += step_rows << 8;
        unsigned FA, FB;                                  // SWAR: flags of pixels 1, 3 in bits 15, 31 of FA and of pixels 0, 2 in bits 15, 31 of FB; exact form: okw[0], okw[1]
        if constexpr (SWAR) {
            // ---- 6-bit form: FOUR pixels per instruction, plain 32-bit and / add / sub (which issue at twice the rate of the packed
            // 16-bit and permute instructions, profiles/r04_valu_rate.txt).  Host-proven superset of the exact test (detect_swar6_threshold);
            // phase 2 recomputes both ring masks exactly, so a few extra survivors cost time, never a result. ----
            constexpr unsigned M6 = 0x3f3f3f3fu;
            unsigned cbk = (unsigned)(128 - g.det_swar_t4) * 0x01010101u;
            asm volatile("" : "+v"(cbk));      // in a VGPR: an SGPR source operand halves the issue rate of the two instructions that use it (profiles/r04_valu_rate.txt)
            const unsigned c6 = (D0 >> 2) & M6, u6 = (Du >> 2) & M6, d6 = (Dd >> 2) & M6;
            const unsigned r6 = __builtin_amdgcn_alignbit(Dp, D0, 26) & M6;      // pixels x + 3: bytes 3 of D0 and 0..2 of Dp, each >> 2
            const unsigned l6 = __builtin_amdgcn_alignbit(D0, Dm, 10) & M6;      // pixels x - 3: bytes 1..3 of Dm and 0 of D0, each >> 2
            const unsigned kb = cbk - c6, kd = cbk + c6;
            const unsigned y4 = r6 + kb, y12 = l6 + kb, y0 = d6 + kb, y8 = u6 + kb;      // bit 7 of a byte: that pixel is brighter than v + th (or nearly)
            const unsigned z4 = kd - r6, z12 = kd - l6, z0 = kd - d6, z8 = kd - u6;      // darker than v - th (or nearly)
            static_assert(COMPASS || !SWAR, "the 6-bit form is a superset filter: only where the arc LUT alone decides (lut_compass)");
            unsigned F = ((y4 | y12) & (y0 | y8)) | ((z4 | z12) & (z0 | z8));
            // rows outside the image's interior: only the first / last step of a band next to the border has one
            unsigned rowmask = cmF;
            if (rbase < ry_lo || rbase + rows_per_step - 1 > ry_hi) {      // a real (scalar) branch: the asm keeps the compiler from turning it into selects on every step
                rowmask = (ry >= ry_lo && ry <= ry_hi) ? cmF : 0u;
                asm volatile("" : "+v"(rowmask));
            }
            F &= rowmask;
            FA = F; FB = F << 8;
        } else {
        const bool row_ok = ry >= ry_lo && ry <= ry_hi;
        unsigned okw[2];                                  // sign bits (15 / 31): pixel 2h / 2h+1 survives both early rejects
#pragma unroll
        for (int h = 0; h < 2; h++) {                     // pixels (0,1) then (2,3); selector byte 0x0c = constant zero
            const s2 v = PK(0u, D0, h ? 0x0c030c02u : 0x0c010c00u);
            const s2 p12 = h ? PK(D0, Dm, 0x0c040c03u) : PK(0u, Dm, 0x0c020c01u);      // x - 3
            const s2 p4 = h ? PK(0u, Dp, 0x0c020c01u) : PK(D0, Dp, 0x0c000c07u);       // x + 3
            const s2 p0 = PK(0u, Dd, h ? 0x0c030c02u : 0x0c010c00u);                   // y + 3
            const s2 p8 = PK(0u, Du, h ? 0x0c030c02u : 0x0c010c00u);                   // y - 3
            // brighter(p) <=> p > v + th, darker(p) <=> p < v - th; an OR over two pixels is a max / min of the pair, an AND over the
            // two pairs a min / max of those: 4 + 2 packed min/max and 2 packed subtractions (sign bits) instead of 16 subtractions
            const s2 vpt = v + th_pk, vmt = v - th_pk;
            const s2 mx_h = __builtin_elementwise_max(p4, p12), mn_h = __builtin_elementwise_min(p4, p12);
            const s2 mx_v = __builtin_elementwise_max(p0, p8), mn_v = __builtin_elementwise_min(p0, p8);
            unsigned ok;
            if (COMPASS) {
                // every mask the arc LUT accepts has two adjacent compass pixels of the same polarity (host-checked property of
                // the LUT): a strict subset of the reference's early rejects that still contains every pixel with a positive score.
                // (B4|B12) & (B0|B8) <=> min(max(p4,p12), max(p0,p8)) > v + th ; (D4|D12) & (D0|D8) <=> max(min, min) < v - th
                const s2 bright = vpt - __builtin_elementwise_min(mx_h, mx_v);
                const s2 dark = __builtin_elementwise_max(mn_h, mn_v) - vmt;
                ok = __builtin_bit_cast(unsigned, (s2)(bright | dark));
            } else {
                // reference: !((near4 && near12) || (near0 && near8)), near(p) <=> |p - v| <= th
                const s2 far_h = (vpt - mx_h) | (mn_h - vmt);                           // sign set: 4 or 12 is far from v
                const s2 far_v = (vpt - mx_v) | (mn_v - vmt);
                ok = __builtin_bit_cast(unsigned, (s2)(far_h & far_v));
            }
            okw[h] = ok & (row_ok ? cm[h] : 0u);
        }
        FA = okw[0]; FB = okw[1];
        }
        // per-wave list append, one ballot per pixel slot: position = n_mine + (survivors of this slot in lower lanes), which
        // v_mbcnt delivers with the base folded in; the survivor count is scalar (s_bcnt1).  Entry order inside the list is free.
        // The flag of slot t is the sign of a 16-bit half: SWAR form - pixel 0 / 2 in FB's low / high half, 1 / 3 in FA's;
        // exact form - pixel 0 / 1 in FA's, 2 / 3 in FB's.
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const unsigned fw = SWAR ? ((t & 1) ? FA : FB) : ((t >> 1) ? FB : FA);
            const bool high = SWAR ? (t >> 1) != 0 : (t & 1) != 0;
            bool keep;
            unsigned long long bal;
            if (HAS_MASK || high) {
                keep = high ? (int)fw < 0 : (fw & 0x8000u) != 0;      // sign of the pixel's 16-bit half
                if (HAS_MASK) { if (keep) keep = mask[(size_t)y * lv.pitch + xs + cb + t] != 0; }
                bal = __ballot(keep);
            } else {
                // the flag is the sign of the LOW half: one 16-bit compare straight into the lane mask (the compiler's own
                // forms - and + compare, bit-field extract + compare, both for one predicate - cost four instructions per slot)
                asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(bal) : "v"(fw));
                keep = __builtin_amdgcn_inverse_ballot_w64(bal);
            }
            // A slot without a survivor (flat image regions: the survivors cluster along edges, two slots in five are empty on the benchmark images) is
            // skipped with ONE scalar branch on the lane mask.  Inside, every lane stores - survivors at the list's tail + their rank, the others into the
            // wave's dump slots: masking the store instead (s_and_saveexec ... s_or exec) costs scalar instructions, and the scalar pipe is as busy as the
            // vector pipes here.  (Measured, pairs/s at C2: exec-masked block with skip branch 118.3 k; no skip, dump slots or masked store 120.6 k;
            // scalar skip + dump slots: this form.)
            if (bal != 0ull) {
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                // (entry = e0 + t as a 32-bit addition: the 16-bit one the compiler picks for the truncated value issues at half the rate)
                unsigned ent = (unsigned)e0;
                if (t) asm("v_add_u32_e32 %0, %1, %2" : "=v"(ent) : "n"(t), "v"(e0));
                unsigned short *const tail = my_list + n_mine;                     // scalar
                *(keep ? tail + pos : dump_ptr) = (unsigned short)ent;
                n_mine += __popcll(bal);
            }
        }
    }
    DET_RING_PASS(true);
#undef DET_RING_PASS
#undef PK
    __syncthreads();
    int n_all = 0;                                        // compact form: positives of the band (pool + spill chunk)
    const unsigned *spill_chunk = nullptr;
    if constexpr (CP) {
        // ---- compact form: the score plane is built now, where the image tile and the survivor lists were (every wave is past them) ----
        // The whole workgroup zeroes the plane with 16-byte stores, a barrier, then thread i scatters entries i, i + 256, ... (pool, then spill chunk).
        // (A form with one barrier less - every wave zeroes the rows it owned in phase 1 and scatters the entries of those rows - was slower: ~30 dword
        // stores per lane instead of 4 wide ones, and every wave reads the whole pool: 110.5 k against 112.5 k pairs/s.)
        {
            const int sstr = L.score_stride;
            const int n16 = (sstr * ((L.score_rows + 1) & ~1) * 2 + 15) >> 4;
            uint4 *z = reinterpret_cast<uint4 *>(s_score);
            for (int i = tid; i < n16; i += DET_THREADS) z[i] = make_uint4(0, 0, 0, 0);
            // (the band's count is read behind the zeroing stores, which do not depend on it: in front of them it was a round trip with the whole workgroup waiting)
            n_all = (int)s_overflow[1];
            if (n_all > pos_cap) spill_chunk = spill + (size_t)(s_overflow[2] - 1u) * (unsigned)chunk_entries;
            __syncthreads();
            // (two loops, not one with a select between the pool and the chunk: a pointer that may be either makes every load a flat_load)
            const int n_lds = min(n_all, pos_cap);
            for (int i = tid; i < n_lds; i += DET_THREADS) {
                const unsigned e = s_pos[i];
                s_score[__umul24((e >> 8) & 255u, (unsigned)sstr) + (e & 255u)] = (unsigned short)(e >> 16);
            }
            if (spill_chunk)
                for (int i = tid; i < n_all - pos_cap; i += DET_THREADS) {
                    const unsigned e = spill_chunk[i];
                    s_score[__umul24((e >> 8) & 255u, (unsigned)sstr) + (e & 255u)] = (unsigned short)(e >> 16);
                }
        }
        __syncthreads();
    }

    // ---- phase 3: 3x3 NMS (>= on the 8 neighbours) + arg-max key, positives of the wave's own list ----
    // (list entries always have a positive score; the `s > 0` test only matters for the dense fallback)
    // K3 picks, per tile, the maximum score; ties go first to the column the horizontal tree prefers, then inside the column to
    // the smaller (ty, k).  When the host has verified that the tree is an arg-max with a fixed column priority (tree_rank_ok),
    // one ds_max_u32 per positive on a per-TILE key (score << 18 | 127 - column priority << 11 | 2047 - row rank) yields the
    // tile winner directly; otherwise the key is per column and phase 4 replays the tree.
    const int SW = L.score_stride;            // (element stride of the plane: score_w, rounded up to even in the compact form)
    const int n_ty = lv.n_ty, recip_nty = lv.recip_nty, recip_tw = lv.recip_tw, recip_th = lv.recip_th;
    int kt_nms = lv.k_tiles;
    asm volatile("" : "+s"(kt_nms));          // opaque: otherwise the compiler re-loads it from the kernel arguments inside the loop below (a scalar memory round trip per iteration)
    const bool ranked = lv.tree_rank_ok != 0;
    const unsigned char *s_rank = reinterpret_cast<const unsigned char *>(s_tree);      // rank[128], inv[128]
    auto nms_one = [&](int ry, int rx, int s_known) {
        if (ry < 1 || ry > th || rx < 1 || rx > ktw) return;          // halo entries only serve as neighbours
        const unsigned short *q = s_score + __umul24(ry, SW) + rx;     // (24-bit multiplies: v_mul_lo_u32 issues at a quarter of the rate)
        const int s = CP ? s_known : q[0];
        if constexpr (CP) {
            // Compact form (the score travels with the entry, the plane is read-only here): the eight neighbours are requested TOGETHER and compared with one
            // maximum - one LDS round trip per positive where the short-circuit chain below is up to eight dependent ones (k_detect -3 %, C2 +0.7 %, tile 58
            // +1.3 %, round 6).  As inline assembly on purpose: written in C++ the compiler merges neighbouring u16 reads into ds_read_b32 at 2-byte aligned
            // addresses, and LDS reads that are not naturally aligned are slow on this part - that form was 3 % SLOWER than the chain, and the ring test's
            // 17 byte reads as 7 unaligned wide ones doubled the kernel's time (profiles/r06_experiments.txt, sections 23-25).
            const unsigned a_mid = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned short *)(q - 1);
            const unsigned a_up = a_mid - 2u * (unsigned)SW, a_dn = a_mid + 2u * (unsigned)SW;
            unsigned n0, n1, n2, n3, n4, n5, n6, n7;
            asm volatile("ds_read_u16 %0, %8\n\tds_read_u16 %1, %8 offset:2\n\tds_read_u16 %2, %8 offset:4\n\t"
                         "ds_read_u16 %3, %9\n\tds_read_u16 %4, %9 offset:4\n\t"
                         "ds_read_u16 %5, %10\n\tds_read_u16 %6, %10 offset:2\n\tds_read_u16 %7, %10 offset:4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5), "=&v"(n6), "=&v"(n7)
                         : "v"(a_up), "v"(a_mid), "v"(a_dn) : "memory");
            const unsigned m = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
            if ((unsigned)s < m) return;
        } else {
            const bool valid = s > 0 && s >= q[-SW - 1] && s >= q[-SW] && s >= q[-SW + 1] && s >= q[-1] && s >= q[1] &&
                               s >= q[SW - 1] && s >= q[SW] && s >= q[SW + 1];      // (full-plane form: all nine reads at once + one maximum + one branch measured 0.9 % slower, rounds 3 and 4)
            if (!valid) return;
        }
        int dy = ry - 1, trow = 0;                                       // tile row inside the band and row inside that tile
        if (R > 1) {
            trow = (int)(__umul24(dy, recip_th) >> 16);                   // dy / th1, exact for dy < 256 (th1 <= 128: dy * (recip * th1 - 65536) < 65536)
            dy -= __umul24(trow, th1);
        }
        const int kk = (int)(__umul24(dy, recip_nty) >> 16), ty = dy - (int)__umul24(kk, n_ty);      // dy / n_ty, exact for dy < 255 (n_ty <= 8; dy * recip < 2^24)
        const unsigned rank = (unsigned)(ty * 256 + kk);                 // lexicographic (ty, k); k < mini_tile <= 128
        if (ranked) {
            const int tile = (int)(__umul24(rx - 1, recip_tw) >> 16), cit = rx - 1 - (int)__umul24(tile, tw);      // (rx-1) / tw, exact for rx-1 < 128 (product < 2^24)
            atomicMax(&s_colkey[__umul24(trow, kt_nms) + tile], ((unsigned)s << 18) | ((127u - s_rank[cit]) << 11) | (2047u - rank));
        } else {
            atomicMax(&s_colkey[rx - 1], ((unsigned)s << 16) | (0xFFFFu - rank));
        }
    };
    if constexpr (CP) {
        const int n_lds = min(n_all, pos_cap);
        for (int i = tid; i < n_lds; i += DET_THREADS) {      // by entry index: every thread of the workgroup takes its share, whichever wave found the positive
            const unsigned e = s_pos[i];
            nms_one((int)((e >> 8) & 255u), (int)(e & 255u), (int)(e >> 16));
        }
        if (spill_chunk)
            for (int i = tid; i < n_all - pos_cap; i += DET_THREADS) {
                const unsigned e = spill_chunk[i];
                nms_one((int)((e >> 8) & 255u), (int)(e & 255u), (int)(e >> 16));
            }
    } else if (!dense) {
        for (int i = lane; i < n_pos; i += 64) {
            const int e = my_list[i];
            nms_one(e >> 8, e & 255, 0);
        }
    } else {
        // the wave's positives did not fit its list (a tile of almost nothing but corners): every pixel of the rows this wave owned
        // in phase 1 is looked at; s_score holds 0 wherever there is no corner
        for (int rbase = wave * rows_per_step; rbase < L.score_rows; rbase += DET_NW * rows_per_step)
            for (int ry = rbase; ry < rbase + rows_per_step && ry < L.score_rows; ry++)
                for (int rx = lane; rx < L.score_w; rx += 64) nms_one(ry, rx, 0);
    }
    __syncthreads();
    if constexpr (CP) {
        // every thread has read its spilled entries (the barrier waits for outstanding loads): the chunk goes back to the arena
        // (a relaxed store: the chunk's next user runs on the same XCD and reaches the chunk through the same L2 as this workgroup's stores - an agent-scope
        // release would write the whole L2 back)
        if (tid == 0 && spill_chunk) __hip_atomic_store(spill_flags + (s_overflow[2] - 1u), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (ranked) {
        // ---- phase 4 (arg-max form): one thread per tile decodes the winner ----
        const int kt = lv.k_tiles;
        int trow = 0;
        for (int rr = 1; rr < R; rr++) trow += tid >= rr * kt;      // R <= DET_MAX_R tile rows of kt tiles
        const int tcol = tid - trow * kt;
        const int tr = tr0 + trow;                        // tile row in the level
        if (tid < R * kt && tr < lv.nth && xg0 + tcol * tw < W) {
            const unsigned key = s_colkey[tid];
            const int sc = (int)(key >> 18);
            const int yt0 = y0 + trow * th1;
            int xx = xg0 + tcol * tw, yy = yt0;           // nothing positive: the tree keeps slot 0's initial value
            if (sc > 0) {
                const int rr = (int)(2047u - (key & 2047u));
                xx += s_rank[128 + (127 - (int)((key >> 11) & 127u))];
                yy = yt0 + (rr >> 8) + (rr & 255) * n_ty;
            }
            const int tile_idx = tr * lv.ntw + grp * kt + tcol;
            tile_out[(size_t)b * g.T + lv.tile_off + tile_idx] =
                ((unsigned long long)(unsigned)sc << 32) | ((unsigned)(yy & 0xFFFF) << 16) | (unsigned)(xx & 0xFFFF);
        }
        return;
    }

    // ---- phase 4: per-tile horizontal tree (literal replay of orb_FAST_apply_NMS_G.cu:1318-1352) ----
    // A slot of the reference's shared array always equals the current register value of its owner thread at a round
    // boundary, so for tiles that fit one wave the tree is replayed with wave shuffles (no LDS traffic, no barriers):
    // a tile occupies a power-of-two lane group (8/16/32/64 lanes), a wave replays 64/group tiles at once.
    if (tw <= 64) {
        // sub = power-of-two lane group that holds one tile; a wave replays 64/sub tiles at once
        const int sub = tw <= 8 ? 8 : tw <= 16 ? 16 : tw <= 32 ? 32 : 64;
        const int tpw = 64 / sub;                         // tiles per wave-step
        const int j = lane & (sub - 1);                   // column inside the tile
        for (int tile0 = wave * tpw; tile0 < lv.k_tiles; tile0 += DET_NW * tpw) {
            const int tile = tile0 + lane / sub;
            const int col = tile * tw + j;
            const bool in_tile = j < tw && tile < lv.k_tiles;
            const bool active = in_tile && (xg0 + col) < W;
            int sc = 0, yy = y0;
            if (in_tile) {
                const unsigned key = s_colkey[col];
                sc = (int)(key >> 16);
                if (sc > 0) {
                    const int rank = (int)(0xFFFFu - (key & 0xFFFFu));
                    yy = y0 + (rank >> 8) + (rank & 255) * lv.n_ty;
                }
            }
            unsigned cur_lo = ((unsigned)(yy & 0xFFFF) << 16) | (unsigned)((xg0 + col) & 0xFFFF);   // pack_kp layout
            int cur_sc = sc;
            int gs = (tw - 1) / 2 + 1;
            for (int it = 0; it < lv.log2_tw; it++) {
                // lane + gs stays inside the lane group whenever j + gs < tw (the only case in which the value is used)
                const int o_sc = __shfl_down(cur_sc, gs, 64);
                const unsigned o_lo = (unsigned)__shfl_down((int)cur_lo, gs, 64);
                if (active && j < gs && j + gs < tw && cur_sc < o_sc) { cur_sc = o_sc; cur_lo = o_lo; }
                gs = (gs - 1) / 2 + 1;
            }
            if (active && j == 0) {
                const int tile_idx = tr0 * lv.ntw + grp * lv.k_tiles + tile;
                tile_out[(size_t)b * g.T + lv.tile_off + tile_idx] = ((unsigned long long)(unsigned)cur_sc << 32) | cur_lo;
            }
        }
        return;
    }
    const bool active = tid < ktw && (xg0 + tid) < W;
    int tile_in_grp = 0, tile_loc = 0;
    unsigned long long cur = 0;
    if (tid < 128) {
        tile_in_grp = tid / tw;
        tile_loc = tid - tile_in_grp * tw;
        int sc = 0, yy = y0;
        if (tid < ktw) {
            const unsigned key = s_colkey[tid];
            sc = (int)(key >> 16);
            if (sc > 0) {
                const int rank = (int)(0xFFFFu - (key & 0xFFFFu));
                yy = y0 + (rank >> 8) + (rank & 255) * lv.n_ty;
            }
        }
        cur = pack_kp(sc, 0, yy, xg0 + tid);
        s_tree[tid] = cur;
    }
    __syncthreads();
    int gs = (tw - 1) / 2 + 1;
    for (int it = 0; it < lv.log2_tw; it++) {
        if (active && tile_loc < gs) {
            if (tile_loc + gs < tw) {
                const unsigned long long t = s_tree[tid + gs];
                if (kp_score(cur) < kp_score(t)) cur = t;
            }
            s_tree[tid] = cur;
        }
        gs = (gs - 1) / 2 + 1;
        __syncthreads();
    }
    if (active && tile_loc == 0) {
        const int tile_idx = tr0 * lv.ntw + grp * lv.k_tiles + tile_in_grp;
        tile_out[(size_t)b * g.T + lv.tile_off + tile_idx] = cur;
    }
}

template <bool HAS_MASK, bool COMPASS, bool SWAR, bool CP>
__global__ __launch_bounds__(DET_THREADS) void k_detect(Geometry g, ImageSrc src, const uint8_t *slab, const uint8_t *mask_slab,
                                                const uint32_t *__restrict__ lut_bits, unsigned long long *tile_out, int n_images,
                                                unsigned *spill, unsigned *spill_flags, int chunk_entries)
{
    // every workgroup-independent kernel argument is pulled into SGPRs by the FIRST round of scalar loads (left alone, the
    // compiler loads each one right before its use, i.e. in 5 dependent rounds before the first image byte can be requested)
    asm volatile("" ::"s"(lut_bits), "s"(slab), "s"(tile_out), "s"(src.l0), "s"(src.l0_stride), "s"(src.l0_pitch), "s"(g.slab_bytes), "s"(g.threshold));
    int b, blk;
    if (!xcd_map(g.detect_blocks, n_images, b, blk)) return;
    detect_workgroup<HAS_MASK, COMPASS, SWAR, CP>(g, src, slab, mask_slab, lut_bits, tile_out, b, blk, spill, spill_flags, chunk_entries);
}

// Single frames: k_blur does not depend on k_detect (both read the pyramid), and a frame is a chain of small launches whose latencies add up -
// so the two run as ONE launch: workgroups [0, detect_blocks) are k_detect's, the rest k_blur's (both 256 threads; LDS = k_detect's dynamic
// part + k_blur's static 10 KB, registers = the larger of the two - irrelevant for one image, which does not fill the chip).  A frame's
// chain is then upload - pyramid - detect+blur - compact - describe: one launch and k_blur's ~8 us less on the critical path.
// (Full-plane form only: single-image handles; a batch handle that is handed one image runs its compact launches.)
static_assert(DET_THREADS == BLUR_THREADS, "the fused launch runs both kinds of workgroups with one block size");
template <bool HAS_MASK, bool COMPASS, bool SWAR>
__global__ __launch_bounds__(DET_THREADS) void k_detect_blur(Geometry g, ImageSrc src, const uint8_t *slab, const uint8_t *mask_slab,
                                                     const uint32_t *__restrict__ lut_bits, unsigned long long *tile_out, uint8_t *blur_slab)
{
    const int blk = (int)blockIdx.x;
    if (blk < g.detect_blocks) detect_workgroup<HAS_MASK, COMPASS, SWAR, false>(g, src, slab, mask_slab, lut_bits, tile_out, 0, blk);
    else blur_workgroup(g, src, slab, blur_slab, lut_bits, 0, blk - g.detect_blocks);
}

#define DETECT_DISPATCH(LAUNCH)                                                                                     \
    do {                                                                                                            \
        if (g.has_mask) { if (!g.lut_compass) LAUNCH(true, false, false); else if (g.det_swar_t4 > 0) LAUNCH(true, true, true); else LAUNCH(true, true, false); }        \
        else            { if (!g.lut_compass) LAUNCH(false, false, false); else if (g.det_swar_t4 > 0) LAUNCH(false, true, true); else LAUNCH(false, true, false); }     \
    } while (0)

// spill / spill_flags: the handle's arena of spill chunks and their busy flags (compact handles; flags zero at creation, every workgroup returns its chunk)
void launch_detect(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *mask_slab,
                   const uint32_t *lut_bits, unsigned long long *tile_out, int n_images, size_t lds_bytes, hipStream_t s, unsigned *spill, unsigned *spill_flags)
{
    if (g.det_compact) {
        const int chunk_entries = detect_spill_chunk_entries(g);
#define DETECT_LAUNCH(M, C, S) hipLaunchKernelGGL((k_detect<M, C, S, true>), xcd_grid(g.detect_blocks, n_images), dim3(DET_THREADS), lds_bytes, s, g, src, slab, mask_slab, lut_bits, tile_out, n_images, spill, spill_flags, chunk_entries)
        DETECT_DISPATCH(DETECT_LAUNCH);
#undef DETECT_LAUNCH
    } else {
#define DETECT_LAUNCH(M, C, S) hipLaunchKernelGGL((k_detect<M, C, S, false>), xcd_grid(g.detect_blocks, n_images), dim3(DET_THREADS), lds_bytes, s, g, src, slab, mask_slab, lut_bits, tile_out, n_images, (unsigned *)nullptr, (unsigned *)nullptr, 0)
        DETECT_DISPATCH(DETECT_LAUNCH);
#undef DETECT_LAUNCH
    }
}

void launch_detect_blur(const Geometry &g, const ImageSrc &src, const uint8_t *slab, const uint8_t *mask_slab, const uint32_t *lut_bits,
                        unsigned long long *tile_out, uint8_t *blur_slab, size_t lds_bytes, hipStream_t s)
{
#define DETECT_LAUNCH(M, C, S) hipLaunchKernelGGL((k_detect_blur<M, C, S>), dim3(g.detect_blocks + g.blur_blocks), dim3(DET_THREADS), lds_bytes, s, g, src, slab, mask_slab, lut_bits, tile_out, blur_slab)
    DETECT_DISPATCH(DETECT_LAUNCH);
#undef DETECT_LAUNCH
}

} // namespace jsorb
