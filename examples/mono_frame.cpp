// mono_frame.cpp - the Frame monocular / RGB-D constructor's front-end half (Frame.cpp:253-330, the same code at :333-420) written
// against the compat shim: ONE ORBExtractor, extract(), two SyncedMem::to_cpu(), SoA unpack - no right image, no stereo match.  The
// frame loop is Tracking's: a Frame is constructed, ASSIGNED to mCurrentFrame and copied to mLastFrame every frame (Tracking.cpp:292,
// 336, 1000), each Frame with SyncedMem members of its own (Frame.h:234-237).
// Usage: mono_frame H W L tile th frames image.raw out.bin
// out.bin: int32 N, kp[6N], desc[32N] of the LAST frame, then mvKeys (N cv::KeyPoint-shaped records)
// Build: g++ -std=c++17 -I include examples/mono_frame.cpp -L jetson_slam_amd -ljsorb -lpthread
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "jsorb_compat.hpp"

namespace {

struct Frame {
    Frame() {}
    // the reference's copy constructor does not copy the SyncedMem members (Frame.cpp:56-78); the implicit copy assignment does
    Frame(const Frame &f) : N(f.N), mvKeys(f.mvKeys), mDescriptors(f.mDescriptors) {}
    Frame &operator=(const Frame &) = default;
    Frame(const unsigned char *imGray, int W, Jetson_SLAM::ORBExtractor *extractor)
    {
        extractor->extract(imGray, W, keypoints_left_, keypoints_desc_left_);      // ExtractORB(0, imGray), Frame.cpp:277
        keypoints_left_.to_cpu();                                                   // Frame.cpp:281-282
        keypoints_desc_left_.to_cpu();
        N = keypoints_left_.count_ / 6;
        mvKeys.resize(N);
        const int *kp = keypoints_left_.cpu_data();                                 // Frame.cpp:286-318
        for (int i = 0; i < N; i++) {
            jsorb_keypoint &k = mvKeys[i];
            k.x = (float)kp[i]; k.y = (float)kp[N + i]; k.response = (float)kp[2 * N + i];
            memcpy(&k.angle, &kp[3 * N + i], 4);
            k.octave = kp[4 * N + i]; k.size = (float)kp[5 * N + i]; k.class_id = -1;
        }
        mDescriptors.assign(keypoints_desc_left_.cpu_data(), keypoints_desc_left_.cpu_data() + (size_t)32 * N);
    }
    int N = 0;
    std::vector<jsorb_keypoint> mvKeys;
    std::vector<unsigned char> mDescriptors;
    orb_cuda::SyncedMem<int> keypoints_left_;
    orb_cuda::SyncedMem<unsigned char> keypoints_desc_left_;
};

} // namespace

int main(int argc, char **argv)
{
    if (argc != 9) { fprintf(stderr, "usage: %s H W L tile th frames image.raw out.bin\n", argv[0]); return 2; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), L = atoi(argv[3]), tile = atoi(argv[4]), th = atoi(argv[5]), frames = atoi(argv[6]);
    std::vector<unsigned char> im((size_t)H * W);
    FILE *f = fopen(argv[7], "rb");
    if (!f || fread(im.data(), 1, im.size(), f) != im.size()) { fprintf(stderr, "cannot read %s\n", argv[7]); return 2; }
    fclose(f);
    try {
        Jetson_SLAM::ORBExtractor ex(H, W, 1.2f, L, 9, 14, 7, th, "", tile, tile, false, false, false, true);
        Frame mCurrentFrame, mLastFrame;
        for (int k = 0; k < frames; k++) {
            if (k & 1) for (size_t i = 0; i < im.size(); i += 97) im[i] ^= 0x10;       // frames of changing content; an even count ends on the original
            mCurrentFrame = Frame(im.data(), W, &ex);                                   // Tracking.cpp:336
            mLastFrame = Frame(mCurrentFrame);                                          // Tracking.cpp:1000
            if (mLastFrame.N != mCurrentFrame.N) { fprintf(stderr, "frame copy lost its keypoints\n"); return 3; }
        }
        // the assigned Frame still owns (shares) the buffers its temporary filled: a real device-to-host copy of them is the frame's keypoints
        std::vector<int> host_side(mCurrentFrame.keypoints_left_.cpu_data_, mCurrentFrame.keypoints_left_.cpu_data_ + mCurrentFrame.keypoints_left_.count_);
        memset(mCurrentFrame.keypoints_left_.cpu_data(), 0xEE, sizeof(int) * mCurrentFrame.keypoints_left_.count_);      // cpu_data(): the host side may have been written
        mCurrentFrame.keypoints_left_.to_cpu();                                                                          // ... so this copies (synced_mem_holder.cpp:88-91)
        if (memcmp(host_side.data(), mCurrentFrame.keypoints_left_.cpu_data_, sizeof(int) * host_side.size())) { fprintf(stderr, "to_cpu() after a host write did not copy\n"); return 3; }
        // a mono flow never asks for a stereo match: the library must not have armed (let alone run) a speculative one
        long adopted = -1, dropped = -1;
        jsorb_speculative_stereo_stats(ex.handle(), &adopted, &dropped);
        if (adopted != 0 || dropped != 0) { fprintf(stderr, "speculative stereo match armed in a mono flow\n"); return 3; }
        std::vector<jsorb_keypoint> mvKeys;
        std::vector<unsigned char> mDescriptors;
        Jetson_SLAM::UnpackFrame(ex, mvKeys, mDescriptors);
        if (mvKeys.size() != (size_t)mCurrentFrame.N || memcmp(mvKeys.data(), mCurrentFrame.mvKeys.data(), mvKeys.size() * sizeof(jsorb_keypoint)) ||
            mDescriptors != mCurrentFrame.mDescriptors) { fprintf(stderr, "UnpackFrame differs from the host loop\n"); return 3; }
        f = fopen(argv[8], "wb");
        const int n = mCurrentFrame.N;
        fwrite(&n, 4, 1, f);
        fwrite(host_side.data(), 4, 6 * (size_t)n, f);
        fwrite(mCurrentFrame.mDescriptors.data(), 1, 32 * (size_t)n, f);
        fwrite(mvKeys.data(), sizeof(jsorb_keypoint), mvKeys.size(), f);
        fclose(f);
        printf("N=%d frames=%d\n", n, frames);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
