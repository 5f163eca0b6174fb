// Prints the constant tables of k_describe (jetson_slam_amd/csrc/describe_tables.h) as JSON - compiled with g++ by tests/test_round3_host_logic.py
#include <cstdio>

#include "describe_tables.h"

int main()
{
    constexpr jsorb::PatternQ q = jsorb::make_pattern_q();
    constexpr jsorb::MomentTab m = jsorb::make_moment_tab();
    static const signed char X[512] = { JSORB_PATTERN_X_VALUES };
    static const signed char Y[512] = { JSORB_PATTERN_Y_VALUES };
    std::printf("{\"pattern_q\": [");
    for (int i = 0; i < 256; i++) std::printf("%s%u", i ? ", " : "", q.v[i]);
    std::printf("], \"slot\": [");
    for (int b = 0; b < 256; b++) std::printf("%s%d", b ? ", " : "", jsorb::pattern_slot(b & 15, b >> 4));
    std::printf("], \"x\": [");
    for (int i = 0; i < 512; i++) std::printf("%s%d", i ? ", " : "", X[i]);
    std::printf("], \"y\": [");
    for (int i = 0; i < 512; i++) std::printf("%s%d", i ? ", " : "", Y[i]);
    std::printf("], \"umax\": [");
    for (int v = 0; v < 16; v++) std::printf("%s%d", v ? ", " : "", jsorb::umax15(v));
    std::printf("], \"moment\": [");
    for (int d = 0; d < 8; d++)
        for (int v = 0; v < 16; v++) std::printf("%s[%u, %u]", (d || v) ? ", " : "", m.v[d][v][0], m.v[d][v][1]);
    std::printf("]}\n");
    return 0;
}
