// describe_tables.h - the constant tables of k_describe.hip as plain constexpr C++17 (no HIP in here): the kernel file includes it,
// and tests/test_round3_host_logic.py compiles tests/cpp/describe_tables_dump.cpp with g++ to check the tables on the CPU (FP8 codes
// against the E4M3 definition, the LDS slot permutation, the moment multipliers against a brute-force sum over the disc).
#pragma once
#include "orb_pattern.inc"

#ifndef JSORB_HALF_PATCH
#define JSORB_HALF_PATCH 15
#endif
#if defined(__HIPCC__) || defined(__CUDACC__)
#define JSORB_HD __host__ __device__
#else
#define JSORB_HD
#endif

namespace jsorb {

// umax[v] for HALF_PATCH 15 (orb_gpu.cpp:161-182 evaluated; checked against the oracle's loop in tests)
JSORB_HD constexpr int umax15(int v)
{
    // {15,15,15,15,14,14,14,13,13,12,11,10,9,8,6,3} packed 4 bits each
    const unsigned long long tab = 0x3689ABCDDEEEFFFFull;
    return (int)((tab >> (4 * v)) & 0xF);
}

// Pattern as FP8 (OCP E4M3: the integers up to 16 are exact, the pattern's coordinates lie in [-13, 13]): one dword per descriptor
// bit = (x0, x1, y0, y1), which two v_cvt_pk_f32_fp8 expand into the register pairs the packed-f32 instructions take.  1 KB of LDS
// instead of 4 KB of floats (one more workgroup per CU: the kernel waits on dependent latencies, so resident waves are what it
// needs) and 4 x 16-byte LDS reads per lane instead of 16.  The 16 dwords of lane sl (steps it = 0..15: descriptor bit it*16 + sl)
// are contiguous; their four 16-byte chunks are rotated by sl >> 2 so that the 16 lanes of a keypoint hit 16 different bank groups.
#ifndef DESC_FP8_BIAS
#define DESC_FP8_BIAS 7
#endif
JSORB_HD constexpr unsigned fp8_e4m3_of_int(int v)
{
    if (v == 0) return 0u;
    const unsigned sgn = v < 0 ? 0x80u : 0u, a = (unsigned)(v < 0 ? -v : v);
    unsigned e = 0;
    while ((a >> (e + 1)) != 0) e++;
    return sgn | ((e + DESC_FP8_BIAS) << 3) | (((a << 3) >> e) & 7u);
}
JSORB_HD constexpr int pattern_slot(int sl, int it) { return sl * 16 + ((((it >> 2) + (sl >> 2)) & 3) << 2) + (it & 3); }
struct PatternQ { unsigned v[256]; };
JSORB_HD constexpr PatternQ make_pattern_q()
{
    constexpr signed char X[512] = { JSORB_PATTERN_X_VALUES };
    constexpr signed char Y[512] = { JSORB_PATTERN_Y_VALUES };
    PatternQ t{};
    for (int b = 0; b < 256; b++)
        t.v[pattern_slot(b & 15, b >> 4)] = fp8_e4m3_of_int(X[2 * b]) | fp8_e4m3_of_int(X[2 * b + 1]) << 8 | fp8_e4m3_of_int(Y[2 * b]) << 16 |
                                            fp8_e4m3_of_int(Y[2 * b + 1]) << 24;
    return t;
}
// Intensity centroid (K8).  The un-blurred patch is staged as 31 rows of 40 B (round 6; rounds 3-5: 48 B at a 56-byte stride) from the 4-byte aligned
// column xa = (x - 15) & ~3.  Lane v (0..15) of a keypoint owns the two rows y + v and y - v, which have the same extent umax[v]:
// in step d (0..7) it takes from each of them the dword of columns u = -15 + 4d .. -12 + 4d (two aligned LDS dwords and a
// v_alignbyte by (x - 15) & 3) and feeds it to v_dot4_u32_u8 with the multipliers of this table: .x = |u| of the four bytes (0
// outside the disc), .y = 1 for a byte inside the disc.  Steps 0..3 hold u <= 0, steps 4..7 u > 0, so the sign of u is a property of
// the unrolled step and the sign of v one of the row: m10 = (P1 + P2) - (N1 + N2), m01 = v * (S1 - S2).  Integer sums: any order
// gives the reference's value.  3 vector instructions per dword instead of 7 (mask, two bit-field extracts, two dots, two
// multiply-adds) and no work list to fetch: the table is 1 KB of LDS per workgroup for every alignment.
struct MomentTab { unsigned v[8][16][2]; };
JSORB_HD constexpr MomentTab make_moment_tab()
{
    MomentTab t{};
    for (int d = 0; d < 8; d++)
        for (int v = 0; v < 16; v++) {
            const int dmax = umax15(v);
            unsigned cu = 0, in = 0;
            for (int j = 0; j < 4; j++) {
                const int u = -JSORB_HALF_PATCH + 4 * d + j, au = u < 0 ? -u : u;
                if (au <= dmax) { cu |= (unsigned)au << (8 * j); in |= 1u << (8 * j); }
            }
            t.v[d][v][0] = cu;
            t.v[d][v][1] = in;
        }
    return t;
}

} // namespace jsorb
