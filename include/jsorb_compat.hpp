// jsorb_compat.hpp - header-only C++ shim that recreates the reference's front-end interface on top of the C ABI
// (include/jsorb.h), so that Frame / Tracking code keeps compiling against the same names:
//
//   orb_cuda::SyncedMem<T>          include/cuda/synced_mem_holder.hpp:10-65   (count_, cpu_data(), gpu_data(), to_cpu(), resize())
//   Jetson_SLAM::ORBExtractor       include/ORBextractor.h:21-93               (ctor argument order, extract(), get_* tables)
//   Jetson_SLAM::ComputeStereoMatches  = body of Frame::ComputeStereoMatches   src/Frame.cpp:780-803
//
// No HIP / OpenCV header is needed by the consumer: device->host copies go through the ABI.  Define JSORB_WITH_OPENCV
// before including to get the cv::Mat overload of extract().
#ifndef JSORB_COMPAT_HPP
#define JSORB_COMPAT_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "jsorb.h"

#ifdef JSORB_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace orb_cuda {

// Result holder with the reference's SyncedMem surface.  The device side is a VIEW of the extractor's result buffers
// (valid until the next extract on that extractor, exactly like the reference, whose extract() resizes and refills the
// caller's SyncedMem); the host side is owned.
template <typename Dtype>
class SyncedMem {
public:
    int count_ = 0;
    const Dtype *gpu_data_ = nullptr;
    std::vector<Dtype> cpu_;

    void resize(int count) { count_ = count; if ((int)cpu_.size() < count) cpu_.resize(count); }
    Dtype *cpu_data() { return cpu_.data(); }
    const Dtype *gpu_data() const { return gpu_data_; }
    // filled by ORBExtractor::extract
    const jsorb_extractor *owner_ = nullptr;
    int image_ = 0;
    void to_cpu();
};

template <>
inline void SyncedMem<int>::to_cpu()
{
    if (owner_ && count_ > 0 && jsorb_copy_keypoints(owner_, image_, cpu_.data()) != JSORB_OK) throw std::runtime_error("jsorb_copy_keypoints failed");
}
template <>
inline void SyncedMem<unsigned char>::to_cpu()
{
    if (owner_ && count_ > 0 && jsorb_copy_descriptors(owner_, image_, cpu_.data()) != JSORB_OK) throw std::runtime_error("jsorb_copy_descriptors failed");
}

} // namespace orb_cuda

namespace Jetson_SLAM {

using orb_cuda::SyncedMem;

class ORBExtractor {
public:
    // argument order of include/ORBextractor.h:25-35; str_mask: "" = no mask.  A mask image has to be decoded by the caller
    // (the reference uses cv::imread) and passed through set-up code as a raw plane via the second constructor.
    ORBExtractor(int im_height, int im_width, float scale_factor, int n_levels, int FAST_N_MIN, int FAST_N_MAX, int th_FAST_MIN,
                 int th_FAST_MAX, std::string str_mask, int tile_h, int tile_w, bool fixed_multi_scale_tile_size, bool apply_nms_ms,
                 bool nms_ms_mode_gpu, bool use_gpu = false)
        : ORBExtractor(im_height, im_width, scale_factor, n_levels, FAST_N_MIN, FAST_N_MAX, th_FAST_MIN, th_FAST_MAX,
                       (const unsigned char *)nullptr, tile_h, tile_w, fixed_multi_scale_tile_size, apply_nms_ms, nms_ms_mode_gpu, use_gpu)
    {
        if (!str_mask.empty()) throw std::invalid_argument("jsorb: pass the decoded mask plane instead of a file name");
    }

    ORBExtractor(int im_height, int im_width, float scale_factor, int n_levels, int FAST_N_MIN, int FAST_N_MAX, int th_FAST_MIN,
                 int th_FAST_MAX, const unsigned char *mask_plane, int tile_h, int tile_w, bool fixed_multi_scale_tile_size,
                 bool apply_nms_ms, bool nms_ms_mode_gpu, bool /*use_gpu*/ = false, int device_id = 0)
    {
        n_levels_ = n_levels;
        scale_factor_ = scale_factor;
        // src/ORBextractor.cpp:43-71
        scale_.resize(n_levels); inv_scale_.resize(n_levels); level_sigma2_.resize(n_levels); inv_level_sigma2_.resize(n_levels);
        scale_[0] = 1.0f; level_sigma2_[0] = 1.0f;
        for (int i = 1; i < n_levels; i++) { scale_[i] = scale_[i - 1] * scale_factor_; level_sigma2_[i] = scale_[i] * scale_[i]; }
        for (int i = 0; i < n_levels; i++) { inv_scale_[i] = 1.0f / scale_[i]; inv_level_sigma2_[i] = 1.0f / level_sigma2_[i]; }
        jsorb_params p{};
        p.height = im_height; p.width = im_width; p.n_levels = n_levels; p.scale_factor = scale_factor;
        p.fast_n_min = FAST_N_MIN; p.fast_n_max = FAST_N_MAX; p.th_fast_min = th_FAST_MIN; p.th_fast_max = th_FAST_MAX;
        p.tile_h = tile_h; p.tile_w = tile_w; p.fixed_multi_scale_tile_size = fixed_multi_scale_tile_size;
        p.apply_nms_ms = apply_nms_ms; p.nms_ms_mode_gpu = nms_ms_mode_gpu; p.device_id = device_id; p.max_batch = 1;
        const int rc = jsorb_create(&p, mask_plane, &orb_gpu_);
        if (rc != JSORB_OK) {
            std::string msg = orb_gpu_ ? jsorb_last_error(orb_gpu_) : "jsorb_create failed";
            if (orb_gpu_) jsorb_destroy(orb_gpu_);
            orb_gpu_ = nullptr;
            throw std::runtime_error("jsorb_create: " + msg);
        }
        width_ = im_width; height_ = im_height;
    }
    ORBExtractor(const ORBExtractor &) = delete;
    ORBExtractor &operator=(const ORBExtractor &) = delete;
    ~ORBExtractor() { if (orb_gpu_) jsorb_destroy(orb_gpu_); }

    // raw-plane form of extract(const cv::Mat&, SyncedMem<int>&, SyncedMem<unsigned char>&)  (ORBextractor.h:40-42)
    void extract(const unsigned char *image, int step, SyncedMem<int> &keypoints, SyncedMem<unsigned char> &keypoints_desc)
    {
        int n = 0;
        if (jsorb_extract(orb_gpu_, image, step, &n) != JSORB_OK) throw std::runtime_error(std::string("jsorb_extract: ") + jsorb_last_error(orb_gpu_));
        keypoints.resize(6 * n); keypoints.gpu_data_ = jsorb_keypoints_device(orb_gpu_, 0); keypoints.owner_ = orb_gpu_; keypoints.image_ = 0;
        keypoints_desc.resize(32 * n); keypoints_desc.gpu_data_ = jsorb_descriptors_device(orb_gpu_, 0); keypoints_desc.owner_ = orb_gpu_; keypoints_desc.image_ = 0;
    }
#ifdef JSORB_WITH_OPENCV
    void extract(const cv::Mat &image, SyncedMem<int> &keypoints, SyncedMem<unsigned char> &keypoints_desc)
    {
        extract(image.data, (int)image.step, keypoints, keypoints_desc);   // explicit step (the reference assumes step == width, orb_gpu.cpp:497)
    }
    void operator()(const cv::Mat &image, SyncedMem<int> &k, SyncedMem<unsigned char> &d) { extract(image, k, d); }
#endif

    int get_levels() { return n_levels_; }
    float get_scale_factor() { return scale_factor_; }
    const std::vector<float> get_scale_factors() { return scale_; }
    std::vector<float> get_inverse_scale_factors() { return inv_scale_; }
    std::vector<float> get_scale_sigma_squares() { return level_sigma2_; }
    std::vector<float> get_inverse_scale_sigma_squares() { return inv_level_sigma2_; }

    jsorb_extractor *orb_gpu_ = nullptr;   // the reference exposes its ORB_GPU* under this name (ORBextractor.h:75)

protected:
    std::vector<float> scale_, inv_scale_, level_sigma2_, inv_level_sigma2_;
    int n_levels_ = 0, width_ = 0, height_ = 0;
    float scale_factor_ = 1.f;
};

// Body of Frame::ComputeStereoMatches (Frame.cpp:780-803): mvuRight / mvDepth sized N_left, -1 = no match.
// TH_HIGH / TH_LOW are ORBmatcher's (ORBmatcher.cpp:24-25).  mb must be mbf/fx (the reference reads an unassigned
// Frame::mb here on a fresh Frame, Frame.cpp:219 vs :247).
inline void ComputeStereoMatches(ORBExtractor &left, ORBExtractor &right, float mb, float mbf, std::vector<float> &mvuRight,
                                 std::vector<float> &mvDepth, jsorb_stereo_stats *stats = nullptr, int TH_HIGH = 100, int TH_LOW = 50)
{
    const int n = jsorb_n_keypoints(left.orb_gpu_, 0);
    if (n < 0) throw std::runtime_error("ComputeStereoMatches before extract");
    mvuRight.assign(n, -1.0f);
    mvDepth.assign(n, -1.0f);
    float dummy = -1.0f;
    const int rc = jsorb_stereo_match(left.orb_gpu_, right.orb_gpu_, mb, mbf, TH_HIGH, TH_LOW, n ? mvuRight.data() : &dummy, n ? mvDepth.data() : &dummy, stats);
    if (rc != JSORB_OK) throw std::runtime_error(std::string("jsorb_stereo_match: ") + jsorb_last_error(left.orb_gpu_));
}

// Body of the keypoint / descriptor unpacking of Frame::Frame (Frame.cpp:119-196) for one image: mvKeys-shaped records (the
// memory layout of cv::KeyPoint) and the N x 32 descriptor rows, produced on the device and fetched with one synchronisation
// instead of SyncedMem::to_cpu() x 2 + a host loop.
inline void UnpackFrame(ORBExtractor &ex, std::vector<jsorb_keypoint> &keys, std::vector<unsigned char> &descriptors)
{
    const int n = jsorb_n_keypoints(ex.orb_gpu_, 0);
    if (n < 0) throw std::runtime_error("UnpackFrame before extract");
    keys.resize(n);
    descriptors.resize((size_t)32 * n);
    if (n && jsorb_unpack_frame(ex.orb_gpu_, 0, keys.data(), descriptors.data()) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_unpack_frame: ") + jsorb_last_error(ex.orb_gpu_));
}
#ifdef JSORB_WITH_OPENCV
inline void UnpackFrame(ORBExtractor &ex, std::vector<cv::KeyPoint> &mvKeys, cv::Mat &mDescriptors)
{
    static_assert(sizeof(cv::KeyPoint) == sizeof(jsorb_keypoint), "jsorb_keypoint mirrors cv::KeyPoint");
    const int n = jsorb_n_keypoints(ex.orb_gpu_, 0);
    if (n < 0) throw std::runtime_error("UnpackFrame before extract");
    mvKeys.resize(n);
    mDescriptors = cv::Mat(n, 32, CV_8UC1);
    if (n && jsorb_unpack_frame(ex.orb_gpu_, 0, reinterpret_cast<jsorb_keypoint *>(mvKeys.data()), mDescriptors.data) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_unpack_frame: ") + jsorb_last_error(ex.orb_gpu_));
}
#endif

// Body of Frame::AssignFeaturesToGrid (Frame.cpp:463-479) for the extracted keypoints (mvKeysUn == mvKeys, rectified stereo):
// mGrid is the reference's std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] (Frame.h:191).
template <std::size_t COLS, std::size_t ROWS>
inline void AssignFeaturesToGrid(ORBExtractor &ex, float mnMinX, float mnMinY, float mfGridElementWidthInv, float mfGridElementHeightInv,
                                 std::vector<std::size_t> (&mGrid)[COLS][ROWS])
{
    const int n = jsorb_n_keypoints(ex.orb_gpu_, 0);
    if (n < 0) throw std::runtime_error("AssignFeaturesToGrid before extract");
    std::vector<int32_t> start(COLS * ROWS + 1), items(n > 0 ? n : 1);
    if (jsorb_assign_features_to_grid(ex.orb_gpu_, 0, mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv, (int)COLS, (int)ROWS,
                                      start.data(), items.data()) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_assign_features_to_grid: ") + jsorb_last_error(ex.orb_gpu_));
    for (std::size_t i = 0; i < COLS; i++)
        for (std::size_t j = 0; j < ROWS; j++) {
            const std::size_t c = i * ROWS + j;
            mGrid[i][j].assign(items.begin() + start[c], items.begin() + start[c + 1]);
        }
}

} // namespace Jetson_SLAM

#endif // JSORB_COMPAT_HPP
