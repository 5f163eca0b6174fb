// Type-check only (g++ -fsyntax-only -DJSORB_WITH_OPENCV against tests/cpp/opencv_double): a Frame-shaped class whose methods use the shim
// exactly the way the reference's Frame / Tracking code uses the headers the shim stands in for.  Written for this test; the call
// shapes follow Frame.cpp:80-250 (stereo ctor), :481-491 (ExtractORB), :780-803 (ComputeStereoMatches) and Tracking.cpp:180-216.
#define JSORB_WITH_OPENCV
#include "jsorb_compat.hpp"

#include <thread>

namespace Jetson_SLAM {

struct ORBmatcher { static const int TH_HIGH = 100, TH_LOW = 50; };

class Frame {
public:
    Frame() {}
    // Frame.cpp:56-78: the reference's user-declared copy constructor (it does not copy the SyncedMem members).  Because it exists, the
    // copy ASSIGNMENT Tracking uses (Tracking.cpp:292/336/364/366: mCurrentFrame = Frame(...)) is the implicit member-wise one - which
    // needs SyncedMem to be copy-assignable.
    Frame(const Frame &frame)
        : mpORBextractorLeft(frame.mpORBextractorLeft), mpORBextractorRight(frame.mpORBextractorRight), mvKeys(frame.mvKeys), mvKeysRight(frame.mvKeysRight),
          mvuRight(frame.mvuRight), mvDepth(frame.mvDepth), mDescriptors(frame.mDescriptors), mbf(frame.mbf), fx(frame.fx), mb(frame.mb), use_gpu_(frame.use_gpu_)
    {
    }
    Frame(const cv::Mat &imLeft, const cv::Mat &imRight, ORBExtractor *extractorLeft, ORBExtractor *extractorRight, float bf_, float fx_)
        : mpORBextractorLeft(extractorLeft), mpORBextractorRight(extractorRight), mbf(bf_), fx(fx_), use_gpu_(true)
    {
        mb = mbf / fx;                                   // moved up from Frame.cpp:247 (the reference reads mb before assigning it)
        std::thread threadLeft(&Frame::ExtractORB, this, 0, imLeft);
        std::thread threadRight(&Frame::ExtractORB, this, 1, imRight);
        threadLeft.join();
        threadRight.join();
        keypoints_left_.to_cpu(); keypoints_right_.to_cpu(); keypoints_desc_left_.to_cpu(); keypoints_desc_right_.to_cpu();
        const int N = keypoints_left_.count_ / 6;
        mvKeys.resize(N);
        const int *kp = keypoints_left_.cpu_data();
        for (int i = 0; i < N; i++) {
            cv::KeyPoint &k = mvKeys[i];
            k.pt.x = kp[i]; k.pt.y = kp[N + i]; k.response = kp[2 * N + i]; k.angle = ((float *)kp)[3 * N + i]; k.octave = kp[4 * N + i]; k.size = kp[5 * N + i];
        }
        mvKeysRight.resize(keypoints_right_.count_ / 6);
        mDescriptors = cv::Mat(N, 32, CV_8UC1);
        memcpy(mDescriptors.data, keypoints_desc_left_.cpu_data(), (size_t)32 * N);
        ComputeStereoMatches();
        UnpackFrame(*mpORBextractorLeft, mvKeys, mDescriptors);      // the device-side replacement of the loop above
    }

    void ExtractORB(int flag, const cv::Mat &im)
    {
        if (flag == 0) mpORBextractorLeft->extract(im, keypoints_left_, keypoints_desc_left_);
        else (*mpORBextractorRight)(im, keypoints_right_, keypoints_desc_right_);
    }

    void ComputeStereoMatches()
    {
        if (use_gpu_) {
            orb_cuda::ORB_GPU &orb_exl = *mpORBextractorLeft->orb_gpu_;
            orb_cuda::ORB_GPU &orb_exr = *mpORBextractorRight->orb_gpu_;
            orb_exl.ORB_compute_stereo_match(ORBmatcher::TH_HIGH, ORBmatcher::TH_LOW, mb, mbf, orb_exl.height_, orb_exl.width_, mvKeys, mvKeysRight, mvuRight,
                                             mvDepth, keypoints_desc_left_.gpu_data(), keypoints_desc_right_.gpu_data(), orb_exl.image_, orb_exr.image_);
        }
    }

    ORBExtractor *mpORBextractorLeft, *mpORBextractorRight;
    SyncedMem<int> keypoints_left_, keypoints_right_;
    SyncedMem<unsigned char> keypoints_desc_left_, keypoints_desc_right_;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    float mbf, fx, mb;
    bool use_gpu_;
};

// Tracking::Tracking (Tracking.cpp:180-216): the extractors are built from the yaml values, the mask as a file name
inline ORBExtractor *make_extractor(int h, int w, const std::string &str_mask)
{
    return new ORBExtractor(h, w, 1.2f, 8, 9, 14, 7, 20, str_mask, 30, 30, false, true, true, true);
}

// Tracking::GrabImageStereo (Tracking.cpp:255-300) and Tracking::Track: a Frame is built, assigned and copied every frame
inline void tracking_like(const cv::Mat &l, const cv::Mat &r, ORBExtractor *exl, ORBExtractor *exr)
{
    Frame mCurrentFrame, mLastFrame;
    mCurrentFrame = Frame(l, r, exl, exr, 47.9f, 435.2f);      // Tracking.cpp:292
    mLastFrame = Frame(mCurrentFrame);                          // Tracking.cpp:1000 (copy construction, then assignment)
    Frame third(mLastFrame);
    (void)third;
}

} // namespace Jetson_SLAM

int main() { return 0; }
