// Per-frame latency of the reference-shaped synchronous API from C++ (no Python in the loop): what Frame::Frame's stereo
// constructor costs per frame.  Usage: frame_latency H W L tile th fx bf left.raw right.raw [frames]
// The raw files may hold SEVERAL images back to back (JSORB_ROTATE_PAIRS=n reads n of them): frame k then works on pair k mod n, so that consecutive
// frames differ - in soak mode (JSORB_CHECK_EVERY_FRAME=1) every frame is compared with what a SECOND pair of handles, which never arms the speculative
// match, computed for its pair before the loop: a speculative match that delivered the previous frame's result would fail here (round-4 review).
// Build: g++ -O2 -std=c++17 -I include tools/micro/frame_latency.cpp -L jetson_slam_amd -ljsorb -lpthread -Wl,-rpath,$PWD/jetson_slam_amd
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "jsorb_compat.hpp"

static std::vector<unsigned char> read_raw(const char *path, size_t n)
{
    std::vector<unsigned char> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 10) { fprintf(stderr, "usage\n"); return 2; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), L = atoi(argv[3]), tile = atoi(argv[4]), th = atoi(argv[5]);
    const float fx = (float)atof(argv[6]), mbf = (float)atof(argv[7]);
    const int frames = argc > 10 ? atoi(argv[10]) : 300;
    const int n_pairs = getenv("JSORB_ROTATE_PAIRS") ? std::max(1, atoi(getenv("JSORB_ROTATE_PAIRS"))) : 1;
    const size_t img_bytes = (size_t)H * W;
    auto imL = read_raw(argv[8], img_bytes * n_pairs), imR = read_raw(argv[9], img_bytes * n_pairs);
    Jetson_SLAM::ORBExtractor exL(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
    Jetson_SLAM::ORBExtractor exR(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
    // The four SyncedMem members of a Frame (Frame.h:234-237).  Default: long-lived objects (what the reference's commented-out static
    // variant would give).  JSORB_FRESH_SYNCEDMEM=1: constructed anew for every frame and destroyed with it, as the shipped Frame does -
    // the shim then patches the frame graph's destinations and serves the buffers from its cache (jsorb_compat.hpp).
    struct FrameMems { orb_cuda::SyncedMem<int> kpL, kpR; orb_cuda::SyncedMem<unsigned char> dL, dR; };
    const bool fresh = getenv("JSORB_FRESH_SYNCEDMEM") != nullptr;
    std::unique_ptr<FrameMems> mems(new FrameMems);
#define kpL mems->kpL
#define kpR mems->kpR
#define dL mems->dL
#define dR mems->dR
    std::vector<float> mvuRight, mvDepth;
    std::vector<jsorb_keypoint> keys, keysR;
    std::vector<unsigned char> desc, descR;
    const bool check_every = getenv("JSORB_CHECK_EVERY_FRAME") != nullptr;
    const bool persistent = getenv("JSORB_PERSISTENT_THREADS") != nullptr;
    std::atomic<int> go{0}, done{0}, cur_pair{0};
    std::atomic<bool> quit{false};
    std::vector<std::thread> workers;
    if (persistent)
        for (int side = 0; side < 2; side++)
            workers.emplace_back([&, side] {
                int seen = 0;
                for (;;) {
                    while (go.load(std::memory_order_acquire) == seen) { if (quit.load()) return; __builtin_ia32_pause(); }
                    seen++;
                    const size_t off = img_bytes * (size_t)cur_pair.load(std::memory_order_relaxed);
                    if (side == 0) exL.extract(imL.data() + off, W, kpL, dL); else exR.extract(imR.data() + off, W, kpR, dR);
                    done.fetch_add(1, std::memory_order_release);
                }
            });
    // soak mode: the expected result of every pair from a second pair of handles through the plain C calls (no speculation is ever armed on them)
    std::vector<std::vector<float>> u0(n_pairs), d0(n_pairs);
    std::vector<std::vector<jsorb_keypoint>> k0(n_pairs);
    if (check_every) {
        Jetson_SLAM::ORBExtractor refL(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
        Jetson_SLAM::ORBExtractor refR(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
        for (int p = 0; p < n_pairs; p++) {
            int nl = 0, nr = 0;
            if (jsorb_extract(refL.handle(), imL.data() + img_bytes * p, W, &nl) != JSORB_OK || jsorb_extract(refR.handle(), imR.data() + img_bytes * p, W, &nr) != JSORB_OK) return 5;
            u0[p].assign(nl, 0.f); d0[p].assign(nl, 0.f);
            jsorb_stereo_stats st;
            if (jsorb_stereo_match(refL.handle(), refR.handle(), mbf / fx, mbf, 100, 50, u0[p].data(), d0[p].data(), &st) != JSORB_OK) return 5;
            std::vector<unsigned char> dd;
            Jetson_SLAM::UnpackFrame(refL, k0[p], dd);
        }
        long a = 0, dr = 0;
        jsorb_speculative_stereo_stats(refL.handle(), &a, &dr);
        if (a != 0) { fprintf(stderr, "the reference handles adopted a speculative match\n"); return 5; }
    }
    double t_ext = 0, t_cpu = 0, t_st = 0, t_unp = 0;
    std::vector<double> per_frame, spawn_us;
    for (int it = -20; it < frames; it++) {
        const int pair = (it + 20) % n_pairs;
        const size_t off = img_bytes * (size_t)pair;
        cur_pair.store(pair, std::memory_order_relaxed);
        const double t0 = now_us();
        if (fresh) mems.reset(new FrameMems);         // inside the timed region: it is part of what a frame costs
        if (persistent) {        // what an integrator gains by keeping the two extractor threads alive (not the reference's code shape)
            done.store(0, std::memory_order_relaxed);
            go.fetch_add(1, std::memory_order_release);
            while (done.load(std::memory_order_acquire) != 2) __builtin_ia32_pause();
        } else {
            std::thread tl([&] { exL.extract(imL.data() + off, W, kpL, dL); });   // Frame.cpp:107-110
            std::thread tr([&] { exR.extract(imR.data() + off, W, kpR, dR); });
            tl.join(); tr.join();
        }
        const double t1 = now_us();
        kpL.to_cpu(); kpR.to_cpu(); dL.to_cpu(); dR.to_cpu();           // Frame.cpp:119-122
        const double t2 = now_us();
        Jetson_SLAM::ComputeStereoMatches(exL, exR, mbf / fx, mbf, mvuRight, mvDepth);
        const double t3 = now_us();
        Jetson_SLAM::UnpackFrame(exL, keys, desc); Jetson_SLAM::UnpackFrame(exR, keysR, descR);      // alternative to the four to_cpu()
        const double t4 = now_us();
        if (check_every) {       // soak mode: every frame must reproduce, bit for bit, what the plain path computed for ITS pair
            const std::vector<float> &ue = u0[pair], &de = d0[pair];
            const std::vector<jsorb_keypoint> &ke = k0[pair];
            if (ue.size() != mvuRight.size() || memcmp(ue.data(), mvuRight.data(), ue.size() * 4) || memcmp(de.data(), mvDepth.data(), de.size() * 4) ||
                ke.size() != keys.size() || memcmp(ke.data(), keys.data(), ke.size() * sizeof(jsorb_keypoint))) {
                fprintf(stderr, "frame %d (pair %d) differs from the plain path's result for that pair\n", it, pair);
                return 4;
            }
        }
        // what the reference's code shape spends on std::thread alone: two no-op threads spawned and joined, timed in the same loop (outside the frame's time)
        const double s0 = now_us();
        { std::thread ta([] {}); std::thread tb([] {}); ta.join(); tb.join(); }
        const double s1 = now_us();
        if (it >= 0) { t_ext += t1 - t0; t_cpu += t2 - t1; t_st += t3 - t2; t_unp += t4 - t3; per_frame.push_back(t3 - t0); spawn_us.push_back(s1 - s0); }
    }
    quit.store(true);
    for (auto &t : workers) t.join();
    // how many of the matches were the ones the library had already enqueued behind the extracts (include/jsorb.h,
    // jsorb_set_speculative_stereo), and: the same frame once more with the feature off must give the same bits
    long adopted = 0, dropped = 0;
    jsorb_speculative_stereo_stats(exL.handle(), &adopted, &dropped);
    std::vector<float> u_spec = mvuRight, d_spec = mvDepth, u_ref, d_ref;
    jsorb_set_speculative_stereo(exL.handle(), 0);
    {
        const size_t off = img_bytes * (size_t)((frames - 1 + 20) % n_pairs);      // the last frame's pair
        exL.extract(imL.data() + off, W, kpL, dL); exR.extract(imR.data() + off, W, kpR, dR);
    }
    Jetson_SLAM::ComputeStereoMatches(exL, exR, mbf / fx, mbf, u_ref, d_ref);
    const bool same = u_ref.size() == u_spec.size() && d_ref.size() == d_spec.size() && !u_ref.empty() &&
                      memcmp(u_ref.data(), u_spec.data(), u_ref.size() * sizeof(float)) == 0 && memcmp(d_ref.data(), d_spec.data(), d_ref.size() * sizeof(float)) == 0;
    if (!same) { fprintf(stderr, "speculative and plain stereo results differ\n"); return 3; }
    std::sort(per_frame.begin(), per_frame.end());
    std::sort(spawn_us.begin(), spawn_us.end());
    const double med = per_frame.empty() ? 0.0 : per_frame[per_frame.size() / 2], p90 = per_frame.empty() ? 0.0 : per_frame[per_frame.size() * 9 / 10];
    const double p10 = per_frame.empty() ? 0.0 : per_frame[per_frame.size() / 10], spawn_med = spawn_us.empty() ? 0.0 : spawn_us[spawn_us.size() / 2];
    if (getenv("JSORB_JSON"))
        printf("{\"frames\": %d, \"pairs_rotated\": %d, \"total_us_median\": %.1f, \"total_us_p10\": %.1f, \"total_us_p90\": %.1f, \"thread_spawn_us\": %.1f, \"extract_lr_us\": %.1f, \"to_cpu_x4_us\": %.1f, \"stereo_us\": %.1f, \"total_us\": %.1f, \"unpack_x2_us\": %.1f, "
               "\"speculative_matches_adopted\": %ld, \"speculative_matches_dropped\": %ld, \"same_bits_without_speculation\": true}\n", frames, n_pairs, med, p10, p90, spawn_med,
               t_ext / frames, t_cpu / frames, t_st / frames, (t_ext + t_cpu + t_st) / frames, t_unp / frames, adopted, dropped);
    else
    printf("per frame (us): extract L||R (2 threads) %.1f, 4x to_cpu %.1f, ComputeStereoMatches %.1f  => %.1f total ; UnpackFrame x2 instead of to_cpu: %.1f ; "
           "speculative matches adopted %ld / dropped %ld ; median %.1f, p10 %.1f, p90 %.1f ; two no-op std::threads spawn + join %.1f\n",
           t_ext / frames, t_cpu / frames, t_st / frames, (t_ext + t_cpu + t_st) / frames, t_unp / frames, adopted, dropped, med, p10, p90, spawn_med);
    return 0;
}
