// stereo_frame.cpp - the Frame stereo constructor's front-end half (Frame.cpp:80-250) written against the compat shim:
// two ORBExtractor objects run from two std::threads, SyncedMem::to_cpu(), SoA unpack, ComputeStereoMatches.
// Usage: stereo_frame H W L tile th fx bf left.raw right.raw out.bin
// out.bin: int32 N_l, N_r, then kp_l[6N_l] desc_l[32N_l] kp_r[6N_r] desc_r[32N_r] uRight[N_l] depth[N_l],
//          then mvKeys (N_l cv::KeyPoint-shaped records, 28 B each), then mGrid of the left image: 64*48 x { int32 count, int32 items[count] }
// Build: g++ -std=c++17 -I include examples/stereo_frame.cpp -L jetson_slam_amd -ljsorb -lpthread
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

int main(int argc, char **argv)
{
    if (argc != 11) { fprintf(stderr, "usage: %s H W L tile th fx bf left.raw right.raw out.bin\n", argv[0]); return 2; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), L = atoi(argv[3]), tile = atoi(argv[4]), th = atoi(argv[5]);
    const float fx = (float)atof(argv[6]), mbf = (float)atof(argv[7]);
    auto imL = read_raw(argv[8], (size_t)H * W), imR = read_raw(argv[9], (size_t)H * W);
    try {
        Jetson_SLAM::ORBExtractor exL(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
        Jetson_SLAM::ORBExtractor exR(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
        orb_cuda::SyncedMem<int> kpL, kpR;
        orb_cuda::SyncedMem<unsigned char> dL, dR;
        std::thread tl([&] { exL.extract(imL.data(), W, kpL, dL); });   // Frame.cpp:107-110
        std::thread tr([&] { exR.extract(imR.data(), W, kpR, dR); });
        tl.join(); tr.join();
        kpL.to_cpu(); kpR.to_cpu(); dL.to_cpu(); dR.to_cpu();           // Frame.cpp:119-122
        const int nl = kpL.count_ / 6, nr = kpR.count_ / 6;
        // Frame::ComputeStereoMatches exactly as the reference writes it (Frame.cpp:780-803): members of orb_cuda::ORB_GPU
        struct Point2f { float x, y; };
        struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };      // stands in for cv::KeyPoint (same members, same layout)
        std::vector<KeyPoint> mvKeysRef(nl), mvKeysRightRef(nr);
        // Frame.cpp:140-160: the SoA of extract() unpacked into cv::KeyPoint (every int converted to float)
        for (int i = 0; i < nl; i++) mvKeysRef[i] = KeyPoint{{(float)kpL.cpu_data_[i], (float)kpL.cpu_data_[nl + i]}, (float)kpL.cpu_data_[5 * nl + i], 0.f, (float)kpL.cpu_data_[2 * nl + i], kpL.cpu_data_[4 * nl + i], -1};
        for (int i = 0; i < nr; i++) mvKeysRightRef[i] = KeyPoint{{(float)kpR.cpu_data_[i], (float)kpR.cpu_data_[nr + i]}, (float)kpR.cpu_data_[5 * nr + i], 0.f, (float)kpR.cpu_data_[2 * nr + i], kpR.cpu_data_[4 * nr + i], -1};
        std::vector<float> mvuRight, mvDepth;
        const float mb = mbf / fx;
        {
            orb_cuda::ORB_GPU &orb_exl = *exL.orb_gpu_;
            orb_cuda::ORB_GPU &orb_exr = *exR.orb_gpu_;
            orb_exl.ORB_compute_stereo_match(100, 50, mb, mbf, orb_exl.height_, orb_exl.width_, mvKeysRef, mvKeysRightRef, mvuRight, mvDepth,
                                             dL.gpu_data(), dR.gpu_data(), orb_exl.image_, orb_exr.image_);
        }
        if (nl > 2) {   // the shim matches the handles' LAST extract: keypoints that are not those (here: two of them swapped) must be refused, not matched silently
            std::vector<KeyPoint> edited(mvKeysRef);
            std::swap(edited[0], edited[nl - 1]);
            std::vector<float> u3, d3;
            bool refused = false;
            try {
                exL.orb_gpu_->ORB_compute_stereo_match(100, 50, mb, mbf, exL.orb_gpu_->height_, exL.orb_gpu_->width_, edited, mvKeysRightRef, u3, d3, dL.gpu_data(),
                                                       dR.gpu_data(), exL.orb_gpu_->image_, exR.orb_gpu_->image_);
            } catch (const std::runtime_error &) { refused = true; }
            if (!refused) { fprintf(stderr, "edited keypoints were matched instead of refused\n"); return 3; }
        }
        {   // the free-function form must give the same bits
            std::vector<float> u2, d2;
            Jetson_SLAM::ComputeStereoMatches(exL, exR, mb, mbf, u2, d2);
            if (u2.size() != mvuRight.size() || memcmp(u2.data(), mvuRight.data(), 4 * u2.size()) || memcmp(d2.data(), mvDepth.data(), 4 * d2.size())) {
                fprintf(stderr, "the two ComputeStereoMatches forms differ\n");
                return 3;
            }
        }
        {   // SyncedMem keeps its own device copy of the results (the reference's extract() writes into the caller's SyncedMem):
            // pull it back with a real device-to-host copy and compare with the host side delivered by extract()
            std::vector<int> host_side(kpL.cpu_data(), kpL.cpu_data() + kpL.count_);
            (void)kpL.gpu_data();                 // invalidates the "host copy is fresh" shortcut
            memset(kpL.cpu_data(), 0xEE, sizeof(int) * kpL.count_);
            kpL.to_cpu_async();
            kpL.sync_stream();
            if (memcmp(host_side.data(), kpL.cpu_data(), sizeof(int) * kpL.count_)) { fprintf(stderr, "SyncedMem device copy differs from host copy\n"); return 3; }
        }
        {   // copies of a SyncedMem (Frame is copied and assigned every frame) share the buffers AND the "host copy is fresh" knowledge: a device-side
            // write through one alias must make to_cpu() of every other alias copy again
            orb_cuda::SyncedMem<int> a;
            a.resize(64);
            for (int i = 0; i < 64; i++) a.cpu_data_[i] = i + 1;
            a.to_gpu();
            a.set_host_fresh(true);                // what ORB_GPU::extract leaves behind
            orb_cuda::SyncedMem<int> b(a);
            b.set_zero_gpu();                      // the alias changes the device side
            a.to_cpu();                            // must not take the shortcut
            for (int i = 0; i < 64; i++)
                if (a.cpu_data_[i] != 0) { fprintf(stderr, "SyncedMem alias: stale host copy after a device-side write through a copy\n"); return 3; }
        }
        FILE *f = fopen(argv[10], "wb");
        fwrite(&nl, 4, 1, f); fwrite(&nr, 4, 1, f);
        fwrite(kpL.cpu_data(), 4, 6 * (size_t)nl, f); fwrite(dL.cpu_data(), 1, 32 * (size_t)nl, f);
        fwrite(kpR.cpu_data(), 4, 6 * (size_t)nr, f); fwrite(dR.cpu_data(), 1, 32 * (size_t)nr, f);
        fwrite(mvuRight.data(), 4, nl, f); fwrite(mvDepth.data(), 4, nl, f);
        // Frame.cpp:119-196 (unpack) and :463-479 (AssignFeaturesToGrid) through the device-side helpers
        std::vector<jsorb_keypoint> mvKeys;
        std::vector<unsigned char> mDescriptors;
        Jetson_SLAM::UnpackFrame(exL, mvKeys, mDescriptors);
        static std::vector<std::size_t> mGrid[64][48];                   // Frame.h:46-47,191
        const float mnMinX = 0.0f, mnMinY = 0.0f, mnMaxX = (float)W, mnMaxY = (float)H;           // Frame.cpp:226-235, no distortion
        Jetson_SLAM::AssignFeaturesToGrid(exL, mnMinX, mnMinY, 64.0f / (mnMaxX - mnMinX), 48.0f / (mnMaxY - mnMinY), mGrid);
        fwrite(mvKeys.data(), sizeof(jsorb_keypoint), mvKeys.size(), f);
        for (int i = 0; i < 64; i++)
            for (int j = 0; j < 48; j++) {
                const int cnt = (int)mGrid[i][j].size();
                fwrite(&cnt, 4, 1, f);
                for (std::size_t v : mGrid[i][j]) { const int iv = (int)v; fwrite(&iv, 4, 1, f); }
            }
        if (mDescriptors.size() != 32 * (size_t)nl || memcmp(mDescriptors.data(), dL.cpu_data(), mDescriptors.size()) != 0) { fprintf(stderr, "UnpackFrame descriptors differ\n"); return 3; }
        fclose(f);
        int matched = 0;
        for (float d : mvDepth) matched += d > 0;
        printf("N_left=%d N_right=%d matched=%d levels=%d\n", nl, nr, matched, exL.get_levels());
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
