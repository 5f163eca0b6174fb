// jsorb_compat.hpp - header-only C++ shim that recreates the reference's GPU-side interface on top of the C ABI (include/jsorb.h),
// so that Frame / Tracking / ORBmatcher keep compiling against the same names.  It stands in for these reference headers:
//
//   include/cuda/synced_mem_holder.hpp:10-65   orb_cuda::SyncedMem<T>     every member: count_, capacity_, cpu_data_, gpu_data_, pitch_,
//                                                                         cu_stream_, resize, resize_pitched, cpu_data, gpu_data,
//                                                                         to_cpu/to_gpu (+count, +async, +stream), sync_stream, set_zero_*
//   include/cuda/orb_gpu.hpp:22-250            orb_cuda::ORB_GPU          ctor argument order, extract(), ORB_compute_stereo_match(),
//                                                                         height_, width_, scale_, inv_scale_, image_ (what Frame.cpp:780-803 touches)
//   include/cuda/orb_matcher.hpp:12-24         orb_cuda::ORB_Search_by_projection_project_on_frame, ORB_compute_distances
//   include/cuda/tracking_gpu.hpp:14-31        tracking_cuda::compute_isInFrustum_GPU
//   include/ORBextractor.h:21-93               Jetson_SLAM::ORBExtractor  ctor argument order, extract(), get_* tables, orb_gpu_
//
// plus three helpers that are the bodies of Frame.cpp:119-196 (UnpackFrame), :463-479 (AssignFeaturesToGrid) and a free-function form of
// Frame::ComputeStereoMatches.  No HIP / CUDA / OpenCV header is needed by the consumer: all device work goes through the ABI.
// Define JSORB_WITH_OPENCV before including to get the cv::Mat / cv::KeyPoint overloads (the exact reference signatures) and mask
// loading through cv::imread; without it masks are decoded by libjsorb itself (PNG, binary PGM / PPM: jsorb_read_mask_image).
#ifndef JSORB_COMPAT_HPP
#define JSORB_COMPAT_HPP

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "jsorb.h"

#ifdef JSORB_WITH_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>
#endif

// The reference's host code names these CUDA types (SyncedMem::cu_stream_, to_cpu_async(cudaStream_t&)); they are opaque here.
#ifndef JSORB_NO_CUDA_TYPEDEFS
typedef void *cudaStream_t;
typedef int cudaError_t;
#endif

namespace orb_cuda {

// Buffer pairs (pinned host + device) released by SyncedMem objects are parked here and handed to the next object that asks for the same
// capacity.  The reference's Frame holds four SyncedMem members (Frame.h:234-237, the static variant is commented out), so EVERY frame
// runs cudaMallocHost + cudaMalloc four times and frees them again; here a steady-state frame allocates nothing.  Per process, bounded,
// guarded by a mutex (the two extractor threads of a stereo frame release / acquire concurrently).
namespace detail {
struct SyncedBufferCache {
    struct Entry { size_t bytes; void *cpu, *gpu; };
    std::mutex mu;
    std::vector<Entry> free_list;
    std::vector<cudaStream_t> free_streams;         // creating and destroying a HIP stream costs milliseconds: the private streams are recycled too
    static SyncedBufferCache &get() { static SyncedBufferCache c; return c; }
    int take_stream(cudaStream_t *st)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!free_streams.empty()) { *st = free_streams.back(); free_streams.pop_back(); return JSORB_OK; }
        }
        return jsorb_mem_stream_create(st);
    }
    void give_stream(cudaStream_t st, bool used)
    {
        if (!st) return;
        if (used) jsorb_mem_stream_sync(st);       // a stream that never carried work needs no wait (hipStreamSynchronize is ~5 us, four per Frame)
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_streams.size() < 64) { free_streams.push_back(st); return; }
        }
        jsorb_mem_stream_destroy(st);
    }
    bool take(size_t bytes, void **cpu, void **gpu)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < free_list.size(); i++)
            if (free_list[i].bytes == bytes) {
                *cpu = free_list[i].cpu; *gpu = free_list[i].gpu;
                free_list[i] = free_list.back(); free_list.pop_back();
                return true;
            }
        return false;
    }
    void give(size_t bytes, void *cpu, void *gpu)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cpu && gpu && free_list.size() < 32) { free_list.push_back(Entry{bytes, cpu, gpu}); return; }
        }
        if (cpu) jsorb_mem_free_host(cpu);
        if (gpu) jsorb_mem_free_device(gpu);
    }
    ~SyncedBufferCache()
    {
        for (auto &e : free_list) { jsorb_mem_free_host(e.cpu); jsorb_mem_free_device(e.gpu); }
        for (auto st : free_streams) jsorb_mem_stream_destroy(st);
    }
};
}

// orb_cuda::SyncedMem<T> (synced_mem_holder.hpp:10-65, synced_mem_holder.cpp:8-199): a pinned host buffer + a device buffer of
// capacity_ elements and a private stream.  Same semantics: resize() grows only and never clears; the to_* calls copy count_
// (or the given count) elements; the *_async forms run on cu_stream_ (or the given stream) and sync_stream() waits for cu_stream_.
// Errors are recorded in cu_error_ and otherwise ignored, as in the reference (every method there is void).
// COPIES: the reference declares no copy operations, so a copied SyncedMem there aliases the original's buffers and stream (and frees
// them twice).  Frame is copied and assigned all the time (Tracking.cpp:292/336/364/366: mCurrentFrame = Frame(...); Frame(mCurrentFrame)),
// so the copy operations exist here, with the reference's aliasing made safe: a copy SHARES the buffers and the stream with the
// original (a reference count decides who frees); an object that has to grow while it shares leaves the old buffers to the others.
template <typename Dtype>
class SyncedMem {
    // shared by every copy of a SyncedMem: the buffers, the private stream, and what is known about them - whether the host side is current
    // (host_fresh: a copy that writes the device side invalidates it for ALL aliases; round-3 review) and whether work the library cannot see
    // may still be using the device buffer (foreign: the caller took gpu_data() or queued a copy on a stream of its own)
    struct Owner {
        void *cpu = nullptr, *gpu = nullptr; cudaStream_t stream = nullptr; size_t bytes = 0; bool pitched = false;
        std::atomic<bool> stream_used{false}, host_fresh{false}, foreign{false};
        std::atomic<int> refs{1};
    };

public:
    SyncedMem() : count_(0), capacity_(0), cpu_data_(nullptr), gpu_data_(nullptr), pitch_(0), cu_stream_(nullptr), cu_error_(0), own_(new Owner)
    {
        cu_error_ = detail::SyncedBufferCache::get().take_stream(&cu_stream_);
        own_->stream = cu_stream_;
    }
    ~SyncedMem() { release(); }
    SyncedMem(const SyncedMem &o)
        : count_(o.count_), capacity_(o.capacity_), cpu_data_(o.cpu_data_), gpu_data_(o.gpu_data_), pitch_(o.pitch_), cu_stream_(o.cu_stream_),
          cu_error_(o.cu_error_), own_(o.own_)
    {
        if (own_) own_->refs.fetch_add(1);
    }
    SyncedMem &operator=(const SyncedMem &o)
    {
        if (this == &o) return *this;
        if (o.own_) o.own_->refs.fetch_add(1);
        release();
        count_ = o.count_; capacity_ = o.capacity_; cpu_data_ = o.cpu_data_; gpu_data_ = o.gpu_data_; pitch_ = o.pitch_; cu_stream_ = o.cu_stream_;
        cu_error_ = o.cu_error_; own_ = o.own_;
        return *this;
    }
    SyncedMem(SyncedMem &&o) noexcept
        : count_(o.count_), capacity_(o.capacity_), cpu_data_(o.cpu_data_), gpu_data_(o.gpu_data_), pitch_(o.pitch_), cu_stream_(o.cu_stream_),
          cu_error_(o.cu_error_), own_(o.own_)
    {
        o.count_ = o.capacity_ = 0; o.cpu_data_ = nullptr; o.gpu_data_ = nullptr; o.cu_stream_ = nullptr; o.own_ = nullptr;
    }
    SyncedMem &operator=(SyncedMem &&o) noexcept
    {
        if (this == &o) return *this;
        release();
        count_ = o.count_; capacity_ = o.capacity_; cpu_data_ = o.cpu_data_; gpu_data_ = o.gpu_data_; pitch_ = o.pitch_; cu_stream_ = o.cu_stream_;
        cu_error_ = o.cu_error_; own_ = o.own_;
        o.count_ = o.capacity_ = 0; o.cpu_data_ = nullptr; o.gpu_data_ = nullptr; o.cu_stream_ = nullptr; o.own_ = nullptr;
        return *this;
    }

    void resize(int count)
    {
        count_ = count;
        set_host_fresh(false);
        if (capacity_ < count_) {
            capacity_ = count_;
            fresh_owner();
            const size_t bytes = (size_t)capacity_ * sizeof(Dtype);
            void *c = nullptr, *g = nullptr;
            if (!detail::SyncedBufferCache::get().take(bytes, &c, &g)) {
                note(jsorb_mem_alloc_host(bytes, &c));
                note(jsorb_mem_alloc_device(bytes, &g));
            }
            cpu_data_ = (Dtype *)c; gpu_data_ = (Dtype *)g;
            own_->cpu = c; own_->gpu = g; own_->bytes = bytes; own_->pitched = false;
        }
    }
    void resize_pitched(size_t width, size_t height)
    {
        count_ = (int)(width * height);
        set_host_fresh(false);
        fresh_owner();
        void *c = nullptr, *g = nullptr;
        note(jsorb_mem_alloc_host((size_t)count_ * sizeof(Dtype), &c));
        note(jsorb_mem_alloc_device_pitched(width * sizeof(Dtype), height, &g, &pitch_));
        cpu_data_ = (Dtype *)c; gpu_data_ = (Dtype *)g;
        own_->cpu = c; own_->gpu = g; own_->bytes = 0; own_->pitched = true;
    }

    // the caller may write through either pointer: the other side is no longer known to be current (to_cpu() then copies, as
    // synced_mem_holder.cpp:88-91 always does)
    Dtype *cpu_data() { set_host_fresh(false); return cpu_data_; }
    Dtype *gpu_data() { set_host_fresh(false); if (own_) own_->foreign.store(true); return gpu_data_; }

    void to_cpu(void) { to_cpu(count_); }
    void to_gpu(void) { to_gpu(count_); }
    void to_cpu(int count)
    {
        if (host_fresh() && count <= count_) return;      // ORBExtractor::extract already delivered the host copy with the device copy and nobody has asked for a pointer since
        note(jsorb_mem_d2h(cpu_data_, gpu_data_, (size_t)count * sizeof(Dtype)));
    }
    void to_gpu(int count) { set_host_fresh(false); note(jsorb_mem_h2d(gpu_data_, cpu_data_, (size_t)count * sizeof(Dtype))); }
    void to_cpu_async(void) { to_cpu_async(cu_stream_, count_); }
    void to_gpu_async(void) { to_gpu_async(cu_stream_, count_); }
    void to_cpu_async(cudaStream_t &cu_stream) { to_cpu_async(cu_stream, count_); }
    void to_gpu_async(cudaStream_t &cu_stream) { to_gpu_async(cu_stream, count_); }
    void to_cpu_async(int count) { to_cpu_async(cu_stream_, count); }
    void to_gpu_async(int count) { to_gpu_async(cu_stream_, count); }
    void to_cpu_async(cudaStream_t &cu_stream, int count)
    {
        if (host_fresh() && count <= count_) return;
        mark_stream(cu_stream);
        note(jsorb_mem_d2h_async(cpu_data_, gpu_data_, (size_t)count * sizeof(Dtype), cu_stream));
    }
    void to_gpu_async(cudaStream_t &cu_stream, int count)
    {
        set_host_fresh(false);
        mark_stream(cu_stream);
        note(jsorb_mem_h2d_async(gpu_data_, cpu_data_, (size_t)count * sizeof(Dtype), cu_stream));
    }
    void sync_stream(void) { note(jsorb_mem_stream_sync(cu_stream_)); }
    void set_zero_gpu(void) { set_host_fresh(false); note(jsorb_mem_set_zero(gpu_data_, (size_t)count_ * sizeof(Dtype))); }
    void set_zero_gpu_async(void) { set_host_fresh(false); mark_stream(cu_stream_); note(jsorb_mem_set_zero_async(gpu_data_, (size_t)count_ * sizeof(Dtype), cu_stream_)); }
    void set_zero_cpu(void) { set_host_fresh(false); if (cpu_data_) memset(cpu_data_, 0, (size_t)count_ * sizeof(Dtype)); }

    // public in the reference ("//private:" is commented out there)
    int count_;
    int capacity_;
    Dtype *cpu_data_;
    Dtype *gpu_data_;
    size_t pitch_;
    cudaStream_t cu_stream_;
    cudaError_t cu_error_;

    // set by ORB_GPU::extract when it fills both sides in one go (the four blocking to_cpu() of Frame.cpp:119-122 then cost nothing);
    // cleared by anything that hands out a pointer or moves data.  The flag lives with the shared buffers: every alias sees it.
    void set_host_fresh(bool v) { if (own_) own_->host_fresh.store(v); }
    bool host_fresh() const { return own_ && own_->host_fresh.load(); }

private:
    Owner *own_;
    void note(int rc) { if (rc != JSORB_OK) cu_error_ = rc; }
    // work enqueued on any stream makes the buffers busy until that stream has run: the release paths wait for the private stream only when it was
    // used (work put on a FOREIGN stream is the caller's to synchronise before the object goes away, as in the reference)
    void mark_stream(cudaStream_t st) { if (!own_) return; if (st == own_->stream) own_->stream_used.store(true); else own_->foreign.store(true); }
    // The reference frees with cudaFree / cudaFreeHost, which wait for the whole device; a recycled buffer must not reach its next user while
    // work the library never saw (kernels on gpu_data(), copies on the caller's streams) may still touch it: one device-wide wait then
    static void settle(Owner *o) { if (o->foreign.load()) { jsorb_mem_buffer_sync(o->gpu); o->foreign.store(false); } }      // (the device that owns the buffer, not the thread's current one)
    static void destroy(Owner *o)
    {
        detail::SyncedBufferCache::get().give_stream(o->stream, o->stream_used.load());      // (waits for the stream's work first, if it ever had any)
        if (o->pitched) { if (o->cpu) jsorb_mem_free_host(o->cpu); if (o->gpu) jsorb_mem_free_device(o->gpu); }
        else if (o->cpu || o->gpu) { settle(o); detail::SyncedBufferCache::get().give(o->bytes, o->cpu, o->gpu); }
        delete o;
    }
    void release()
    {
        if (own_ && own_->refs.fetch_sub(1) == 1) destroy(own_);
        own_ = nullptr; cpu_data_ = nullptr; gpu_data_ = nullptr; cu_stream_ = nullptr;
    }
    // before new buffers are attached: sole owner -> the old ones go back to the cache; shared -> they stay with the other objects and
    // this one continues with an owner (and stream) of its own
    void fresh_owner()
    {
        if (own_ && own_->refs.load() == 1) {
            if (own_->pitched) { if (own_->cpu) jsorb_mem_free_host(own_->cpu); if (own_->gpu) jsorb_mem_free_device(own_->gpu); }
            else if (own_->cpu || own_->gpu) {
                if (own_->stream && own_->stream_used.load()) jsorb_mem_stream_sync(own_->stream);
                settle(own_);
                detail::SyncedBufferCache::get().give(own_->bytes, own_->cpu, own_->gpu);
            }
            own_->cpu = own_->gpu = nullptr;
            own_->host_fresh.store(false);
        } else {
            if (own_) own_->refs.fetch_sub(1);
            own_ = new Owner;
            cu_stream_ = nullptr;
            note(detail::SyncedBufferCache::get().take_stream(&cu_stream_));
            own_->stream = cu_stream_;
        }
        cpu_data_ = nullptr; gpu_data_ = nullptr;
    }
};

// orb_cuda::ORB_Search_by_projection_project_on_frame (orb_matcher.hpp:12-18, orb_matcher.cu:62-89): synchronous, device pointers
inline void ORB_Search_by_projection_project_on_frame(int n_points, float *Px_gpu, float *Py_gpu, float *Pz_gpu, float *Rcw_gpu, float *tcw_gpu,
                                                      float &fx, float &fy, float &cx, float &cy, float &minX, float &maxX, float &minY, float &maxY,
                                                      float *u_gpu, float *v_gpu, float *invz_gpu, unsigned char *is_valid_gpu)
{
    if (jsorb_project_points(nullptr, n_points, Px_gpu, Py_gpu, Pz_gpu, Rcw_gpu, tcw_gpu, fx, fy, cx, cy, minX, maxX, minY, maxY, u_gpu, v_gpu, invz_gpu,
                             is_valid_gpu) != JSORB_OK)
        throw std::runtime_error("jsorb_project_points failed");
}
// orb_cuda::ORB_compute_distances (orb_matcher.hpp:20-24, orb_matcher.cu:122-144)
inline void ORB_compute_distances(int n_points, int *idx_left, int *idx_right, unsigned char *descriptor_left, unsigned char *descriptor_right, int *distance)
{
    if (jsorb_hamming_pairs(nullptr, n_points, idx_left, idx_right, descriptor_left, descriptor_right, distance) != JSORB_OK)
        throw std::runtime_error("jsorb_hamming_pairs failed");
}


// orb_cuda::ORB_GPU (include/cuda/orb_gpu.hpp:22-250, src/cuda/orb_gpu.cpp): owns one jsorb handle.  Only the members the reference's
// untouched host code reaches are recreated.
class ORB_GPU {
public:
    // what Frame::ComputeStereoMatches passes as orb_exl.image_ / orb_exr.image_ (Frame.cpp:799-800): the un-blurred pyramid of the last
    // extract, which lives inside the handle
    struct ImagePyramid {
        ORB_GPU *owner = nullptr;
        size_t size() const { return owner ? (size_t)jsorb_n_levels(owner->handle_) : 0; }
        const unsigned char *gpu_data(int level) const { return jsorb_level_image_device(owner->handle_, 0, level, 0); }
    };

    // argument order of orb_gpu.hpp:26-36
    ORB_GPU(int im_height, int im_width, int n_levels, float scale_factor, int FAST_N_MIN, int FAST_N_MAX, int th_FAST_MIN, int th_FAST_MAX, int tile_h,
            int tile_w, bool fixed_multi_scale_tile_size, bool apply_nms_ms, bool nms_ms_mode_gpu, std::string str_mask, int device_id = 0,
            const unsigned char *mask_plane = nullptr, int max_batch = 1)
    {
        std::vector<unsigned char> mask;
        int mask_w = im_width, mask_h = im_height;
        if (!mask_plane && !str_mask.empty()) {
            // orb_gpu.cpp:64-75: cv::imread(str_mask); an unreadable file means "no mask" there (mask.empty() -> all 255)
            int mw = 0, mh = 0;
            bool have = false;
#ifdef JSORB_WITH_OPENCV
            cv::Mat m = cv::imread(str_mask);
            if (!m.empty()) {
                cv::cvtColor(m, m, cv::COLOR_BGR2GRAY);
                mw = m.cols; mh = m.rows;
                mask.resize((size_t)mw * mh);
                for (int y = 0; y < mh; y++) memcpy(&mask[(size_t)y * mw], m.ptr(y), mw);
                have = true;
            }
#else
            // without OpenCV: libjsorb's own decoder (PNG, binary PGM / PPM) with cvtColor(BGR2GRAY)'s 8-bit arithmetic
            int rc = jsorb_read_mask_image(str_mask.c_str(), &mw, &mh, nullptr, 0);
            if (rc == JSORB_OK) {
                mask.resize((size_t)mw * mh);
                rc = jsorb_read_mask_image(str_mask.c_str(), &mw, &mh, mask.data(), mask.size());
            }
            if (rc == JSORB_OK) have = true;
            else if (rc != JSORB_ERR_STATE) throw std::invalid_argument(std::string("jsorb: mask file: ") + jsorb_mask_image_last_error());
#endif
            // the reference resizes whatever size the mask has to EVERY level (level 0 included) with INTER_NN (orb_gpu.cpp:77-81): the mask goes
            // through the ABI at its own size and the library resizes each level from it directly
            if (have) { mask_plane = mask.data(); mask_w = mw; mask_h = mh; }
        }
        jsorb_params p{};
        p.height = im_height; p.width = im_width; p.n_levels = n_levels; p.scale_factor = scale_factor;
        p.fast_n_min = FAST_N_MIN; p.fast_n_max = FAST_N_MAX; p.th_fast_min = th_FAST_MIN; p.th_fast_max = th_FAST_MAX;
        p.tile_h = tile_h; p.tile_w = tile_w; p.fixed_multi_scale_tile_size = fixed_multi_scale_tile_size;
        p.apply_nms_ms = apply_nms_ms; p.nms_ms_mode_gpu = nms_ms_mode_gpu; p.device_id = device_id; p.max_batch = max_batch;
        const int rc = jsorb_create_masked(&p, mask_plane, mask_w, mask_h, &handle_);
        if (rc != JSORB_OK) {
            std::string msg = handle_ ? jsorb_last_error(handle_) : "jsorb_create failed";
            if (handle_) jsorb_destroy(handle_);
            handle_ = nullptr;
            throw std::runtime_error("jsorb_create: " + msg);
        }
        n_levels_ = n_levels;
        for (int i = 0; i < n_levels; i++) {
            int h = 0, w = 0;
            jsorb_level_dims(handle_, i, &h, &w, nullptr);
            height_.push_back(h); width_.push_back(w);
            scale_.push_back(jsorb_scale(handle_, i)); inv_scale_.push_back(jsorb_inv_scale(handle_, i));
        }
        image_.owner = this;
    }
    ORB_GPU(const ORB_GPU &) = delete;
    ORB_GPU &operator=(const ORB_GPU &) = delete;
    ~ORB_GPU() { if (handle_) jsorb_destroy(handle_); }

    // ORB_GPU::extract (orb_gpu.hpp:40-42, orb_gpu.cpp:489-841), raw-plane form with an explicit row step (the reference assumes
    // step == width, orb_gpu.cpp:497).  The caller's SyncedMems are resized by the callee and receive the results on BOTH sides: the
    // device side by a device-to-device copy on the handle's stream, the host side from the handle's pinned mirror.
    void extract(const unsigned char *image, int step, SyncedMem<int> &out_keypoints, SyncedMem<unsigned char> &out_keypoints_desc)
    {
        // the caller's SyncedMems are sized once for the keypoint cap T (resize grows only), so that the handle can deliver the device
        // copies in the same stream round trip as the counts; count_ is then set to the frame's 6N / 32N
        const int T = jsorb_total_tiles(handle_);
        if (out_keypoints.capacity_ < 6 * T) out_keypoints.resize(6 * T);
        if (out_keypoints_desc.capacity_ < 32 * T) out_keypoints_desc.resize(32 * T);
        int n = 0;
        if (jsorb_extract_into(handle_, image, step, &n, out_keypoints.gpu_data_, out_keypoints_desc.gpu_data_) != JSORB_OK)
            throw std::runtime_error(std::string("jsorb_extract: ") + jsorb_last_error(handle_));
        out_keypoints.resize(6 * n);
        out_keypoints_desc.resize(32 * n);
        if (n > 0 && (jsorb_copy_keypoints(handle_, 0, out_keypoints.cpu_data_) != JSORB_OK ||             // host side: from the handle's pinned mirror
                      jsorb_copy_descriptors(handle_, 0, out_keypoints_desc.cpu_data_) != JSORB_OK))
            throw std::runtime_error("jsorb: delivering the extract results failed");
        out_keypoints.set_host_fresh(true); out_keypoints_desc.set_host_fresh(true);
    }
#ifdef JSORB_WITH_OPENCV
    void extract(const cv::Mat &image, SyncedMem<int> &out_keypoints, SyncedMem<unsigned char> &out_keypoints_desc)
    {
        extract(image.data, (int)image.step, out_keypoints, out_keypoints_desc);
    }
#endif

    // ORB_GPU::ORB_compute_stereo_match (orb_gpu.hpp:218-229, orb_stereo_match.cu:105-580) with the reference's parameter list, so
    // that Frame::ComputeStereoMatches (Frame.cpp:780-803) compiles unchanged.  It matches the LAST extract of the two handles - which
    // is what mvKeys / mvKeysRight / the descriptor pointers / the two pyramids describe at that call site; the keypoint vectors are
    // compared with the handles' own keypoints (below), the descriptor pointers are not read.  KeyPoint is cv::KeyPoint in the reference.
    template <class KeyPoint>
    void ORB_compute_stereo_match(int ORB_TH_HIGH, int ORB_TH_LOW, float mb, float mbf, std::vector<int> & /*octave_height*/, std::vector<int> & /*octave_width*/,
                                  std::vector<KeyPoint> &mvKeys, std::vector<KeyPoint> &mvKeysRight, std::vector<float> &mvuRight, std::vector<float> &mvDepth,
                                  unsigned char * /*descriptor_left_gpu*/, unsigned char * /*descriptor_right_gpu*/, ImagePyramid &images_left,
                                  ImagePyramid &images_right)
    {
        if (images_left.owner != this || !images_right.owner) throw std::invalid_argument("ORB_compute_stereo_match: pyramids do not belong to these extractors");
        jsorb_extractor *l = handle_, *r = images_right.owner->handle_;
        // Frame's stereo call shape (two extracts, then this call, every frame): ask the library to run the NEXT frame's match right behind
        // its two extracts on the GPU (jsorb.h, jsorb_set_speculative_stereo).  Opt-in here, not a library default: a mono / RGB-D
        // flow never reaches this function.
        if (!speculation_requested_) { jsorb_set_speculative_stereo(l, 1); speculation_requested_ = true; }
        const int n = jsorb_n_keypoints(l, 0);
        if (n < 0 || (size_t)n != mvKeys.size() || (size_t)jsorb_n_keypoints(r, 0) != mvKeysRight.size())
            throw std::runtime_error("ORB_compute_stereo_match: keypoint vectors do not match the last extract of the handles");
        // The reference READS mvKeys / mvKeysRight (orb_stereo_match.cu:119-184); this implementation matches what the handles extracted last.  A
        // caller that filtered, reordered or edited its keypoints in between would get silently different results, so the keypoint
        // vectors are compared with the handles' own (position and octave): all of them on the first call and in debug builds, a strided
        // sample afterwards.  JSORB_COMPAT_NO_KEYPOINT_CHECK removes the check (two device-to-host copies of 24 N bytes per call).
#ifndef JSORB_COMPAT_NO_KEYPOINT_CHECK
        {
#ifdef NDEBUG
            const size_t stride = keypoints_checked_ ? 16 : 1;
#else
            const size_t stride = 1;
#endif
            auto same = [&](jsorb_extractor *h, const std::vector<KeyPoint> &keys, std::vector<int32_t> &soa) {
                const size_t m = keys.size();
                if (!m) return true;
                soa.resize(6 * m);
                if (jsorb_copy_keypoints(h, 0, soa.data()) != JSORB_OK) return false;
                for (size_t i = 0; i < m; i += (i + stride < m || i + 1 == m) ? stride : m - 1 - i)      // the strided sample always includes the last one
                    if (keys[i].pt.x != (float)soa[i] || keys[i].pt.y != (float)soa[m + i] || keys[i].octave != soa[4 * m + i]) return false;
                return true;
            };
            if (!same(l, mvKeys, check_soa_) || !same(r, mvKeysRight, check_soa_))
                throw std::runtime_error("ORB_compute_stereo_match: mvKeys / mvKeysRight are not the keypoints of the handles' last extract (filtered or "
                                         "reordered?) - this implementation matches the last extract, see INTEGRATION.md");
            keypoints_checked_ = true;
        }
#endif
        mvuRight.resize(mvKeys.size(), -1.0f);          // orb_stereo_match.cu:496-497 (resize keeps earlier contents; the ABI call overwrites all n)
        mvDepth.resize(mvKeys.size(), -1.0f);
        float dummy = -1.0f;
        if (jsorb_stereo_match(l, r, mb, mbf, ORB_TH_HIGH, ORB_TH_LOW, n ? mvuRight.data() : &dummy, n ? mvDepth.data() : &dummy, &last_stereo_stats_) != JSORB_OK)
            throw std::runtime_error(std::string("jsorb_stereo_match: ") + jsorb_last_error(l));
    }

    jsorb_extractor *handle() const { return handle_; }

    // public data members of the reference that host code outside ORB_GPU reads (orb_gpu.hpp:240-250; Frame.cpp:784-801)
    int n_levels_ = 0;
    std::vector<int> height_, width_;
    std::vector<float> scale_, inv_scale_;
    ImagePyramid image_;
    jsorb_stereo_stats last_stereo_stats_{};
    bool speculation_requested_ = false;
    bool keypoints_checked_ = false;
    std::vector<int32_t> check_soa_;

private:
    jsorb_extractor *handle_ = nullptr;
};

} // namespace orb_cuda

namespace tracking_cuda {
// tracking_cuda::compute_isInFrustum_GPU (tracking_gpu.hpp:14-31, tracking_isinfrustum.cu:19-160): synchronous, device pointers
inline void compute_isInFrustum_GPU(int n_points, float *Px_gpu, float *Py_gpu, float *Pz_gpu, float *Pnx_gpu, float *Pny_gpu, float *Pnz_gpu,
                                    float *MaxDistance_gpu, float *invariance_maxDistance_gpu, float *invariance_minDistance_gpu, float *Rcw_gpu,
                                    float *tcw_gpu, float *Ow_gpu, float &fx, float &fy, float &cx, float &cy, int &minX, int &maxX, int &minY, int &maxY,
                                    int &nScaleLevels, float &logScaleFactor, float &viewCosAngle, float *invz_gpu, float *u_gpu, float *v_gpu,
                                    int *predictedlevel_gpu, float *viewCos_gpu, unsigned char *is_infrustum_gpu)
{
    if (jsorb_is_in_frustum(nullptr, n_points, Px_gpu, Py_gpu, Pz_gpu, Pnx_gpu, Pny_gpu, Pnz_gpu, MaxDistance_gpu, invariance_maxDistance_gpu,
                            invariance_minDistance_gpu, Rcw_gpu, tcw_gpu, Ow_gpu, fx, fy, cx, cy, minX, maxX, minY, maxY, nScaleLevels, logScaleFactor,
                            viewCosAngle, invz_gpu, u_gpu, v_gpu, predictedlevel_gpu, viewCos_gpu, is_infrustum_gpu) != JSORB_OK)
        throw std::runtime_error("jsorb_is_in_frustum failed");
}
} // namespace tracking_cuda

namespace Jetson_SLAM {

using orb_cuda::SyncedMem;

class ORBExtractor {
public:
    // argument order of include/ORBextractor.h:25-35; str_mask: "" = no mask, else an image file (see ORB_GPU above)
    ORBExtractor(int im_height, int im_width, float scale_factor, int n_levels, int FAST_N_MIN, int FAST_N_MAX, int th_FAST_MIN,
                 int th_FAST_MAX, std::string str_mask, int tile_h, int tile_w, bool fixed_multi_scale_tile_size, bool apply_nms_ms,
                 bool nms_ms_mode_gpu, bool use_gpu = false)
    {
        init(im_height, im_width, scale_factor, n_levels, use_gpu);
        // src/ORBextractor.cpp:75-87 (the GPU object is built whatever use_gpu says; device 0)
        orb_gpu_ = new orb_cuda::ORB_GPU(im_height, im_width, n_levels, scale_factor, FAST_N_MIN, FAST_N_MAX, th_FAST_MIN, th_FAST_MAX, tile_h, tile_w,
                                         fixed_multi_scale_tile_size, apply_nms_ms, nms_ms_mode_gpu, str_mask, 0);
    }
    // additions: the mask as a decoded level-0 plane (NULL = none), an explicit device, a batch capacity
    ORBExtractor(int im_height, int im_width, float scale_factor, int n_levels, int FAST_N_MIN, int FAST_N_MAX, int th_FAST_MIN,
                 int th_FAST_MAX, const unsigned char *mask_plane, int tile_h, int tile_w, bool fixed_multi_scale_tile_size,
                 bool apply_nms_ms, bool nms_ms_mode_gpu, bool use_gpu = false, int device_id = 0, int max_batch = 1)
    {
        init(im_height, im_width, scale_factor, n_levels, use_gpu);
        orb_gpu_ = new orb_cuda::ORB_GPU(im_height, im_width, n_levels, scale_factor, FAST_N_MIN, FAST_N_MAX, th_FAST_MIN, th_FAST_MAX, tile_h, tile_w,
                                         fixed_multi_scale_tile_size, apply_nms_ms, nms_ms_mode_gpu, std::string(), device_id, mask_plane, max_batch);
    }
    ORBExtractor(const ORBExtractor &) = delete;
    ORBExtractor &operator=(const ORBExtractor &) = delete;
    ~ORBExtractor() { delete orb_gpu_; }

    // raw-plane form of extract(const cv::Mat&, SyncedMem<int>&, SyncedMem<unsigned char>&)  (ORBextractor.h:40-42)
    void extract(const unsigned char *image, int step, SyncedMem<int> &keypoints, SyncedMem<unsigned char> &keypoints_desc)
    {
        orb_gpu_->extract(image, step, keypoints, keypoints_desc);
    }
#ifdef JSORB_WITH_OPENCV
    void extract(const cv::Mat &image, SyncedMem<int> &keypoints, SyncedMem<unsigned char> &keypoints_desc) { orb_gpu_->extract(image, keypoints, keypoints_desc); }
    void operator()(const cv::Mat &image, SyncedMem<int> &k, SyncedMem<unsigned char> &d) { extract(image, k, d); }
#endif

    int get_levels() { return n_levels_; }
    float get_scale_factor() { return scale_factor_; }
    const std::vector<float> get_scale_factors() { return scale_; }
    std::vector<float> get_inverse_scale_factors() { return inv_scale_; }
    std::vector<float> get_scale_sigma_squares() { return level_sigma2_; }
    std::vector<float> get_inverse_scale_sigma_squares() { return inv_level_sigma2_; }

    orb_cuda::ORB_GPU *orb_gpu_ = nullptr;   // ORBextractor.h:75
    jsorb_extractor *handle() const { return orb_gpu_->handle(); }

protected:
    void init(int im_height, int im_width, float scale_factor, int n_levels, bool use_gpu)
    {
        n_levels_ = n_levels;
        scale_factor_ = scale_factor;
        use_gpu_ = use_gpu;
        // src/ORBextractor.cpp:43-71
        scale_.resize(n_levels); inv_scale_.resize(n_levels); level_sigma2_.resize(n_levels); inv_level_sigma2_.resize(n_levels);
        scale_[0] = 1.0f; level_sigma2_[0] = 1.0f;
        for (int i = 1; i < n_levels; i++) { scale_[i] = scale_[i - 1] * scale_factor_; level_sigma2_[i] = scale_[i] * scale_[i]; }
        for (int i = 0; i < n_levels; i++) { inv_scale_[i] = 1.0f / scale_[i]; inv_level_sigma2_[i] = 1.0f / level_sigma2_[i]; }
        width_.assign(1, im_width); height_.assign(1, im_height);
    }
    std::vector<int> height_, width_;
    std::vector<float> scale_, inv_scale_, level_sigma2_, inv_level_sigma2_;
    int n_levels_ = 0;
    float scale_factor_ = 1.f;
    bool use_gpu_ = false;
};

// Free-function form of Frame::ComputeStereoMatches (Frame.cpp:780-803): mvuRight / mvDepth sized N_left, -1 = no match.
// TH_HIGH / TH_LOW are ORBmatcher's (ORBmatcher.cpp:24-25).  mb must be mbf/fx (the reference reads an unassigned Frame::mb here on a
// fresh Frame, Frame.cpp:219 vs :247).
inline void ComputeStereoMatches(ORBExtractor &left, ORBExtractor &right, float mb, float mbf, std::vector<float> &mvuRight,
                                 std::vector<float> &mvDepth, jsorb_stereo_stats *stats = nullptr, int TH_HIGH = 100, int TH_LOW = 50)
{
    const int n = jsorb_n_keypoints(left.handle(), 0);
    if (n < 0) throw std::runtime_error("ComputeStereoMatches before extract");
    if (!left.orb_gpu_->speculation_requested_) { jsorb_set_speculative_stereo(left.handle(), 1); left.orb_gpu_->speculation_requested_ = true; }
    mvuRight.assign(n, -1.0f);
    mvDepth.assign(n, -1.0f);
    float dummy = -1.0f;
    const int rc = jsorb_stereo_match(left.handle(), right.handle(), mb, mbf, TH_HIGH, TH_LOW, n ? mvuRight.data() : &dummy, n ? mvDepth.data() : &dummy, stats);
    if (rc != JSORB_OK) throw std::runtime_error(std::string("jsorb_stereo_match: ") + jsorb_last_error(left.handle()));
}

// Body of the keypoint / descriptor unpacking of Frame::Frame (Frame.cpp:119-196) for one image: mvKeys-shaped records (the
// memory layout of cv::KeyPoint) and the N x 32 descriptor rows.
inline void UnpackFrame(ORBExtractor &ex, std::vector<jsorb_keypoint> &keys, std::vector<unsigned char> &descriptors)
{
    const int n = jsorb_n_keypoints(ex.handle(), 0);
    if (n < 0) throw std::runtime_error("UnpackFrame before extract");
    keys.resize(n);
    descriptors.resize((size_t)32 * n);
    if (n && jsorb_unpack_frame(ex.handle(), 0, keys.data(), descriptors.data()) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_unpack_frame: ") + jsorb_last_error(ex.handle()));
}
#ifdef JSORB_WITH_OPENCV
inline void UnpackFrame(ORBExtractor &ex, std::vector<cv::KeyPoint> &mvKeys, cv::Mat &mDescriptors)
{
    static_assert(sizeof(cv::KeyPoint) == sizeof(jsorb_keypoint), "jsorb_keypoint mirrors cv::KeyPoint");
    const int n = jsorb_n_keypoints(ex.handle(), 0);
    if (n < 0) throw std::runtime_error("UnpackFrame before extract");
    mvKeys.resize(n);
    mDescriptors = cv::Mat(n, 32, CV_8UC1);
    if (n && jsorb_unpack_frame(ex.handle(), 0, reinterpret_cast<jsorb_keypoint *>(mvKeys.data()), mDescriptors.data) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_unpack_frame: ") + jsorb_last_error(ex.handle()));
}
#endif

// Body of Frame::AssignFeaturesToGrid (Frame.cpp:463-479) for the extracted keypoints (mvKeysUn == mvKeys, rectified stereo):
// mGrid is the reference's std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS] (Frame.h:191).
template <std::size_t COLS, std::size_t ROWS>
inline void AssignFeaturesToGrid(ORBExtractor &ex, float mnMinX, float mnMinY, float mfGridElementWidthInv, float mfGridElementHeightInv,
                                 std::vector<std::size_t> (&mGrid)[COLS][ROWS])
{
    const int n = jsorb_n_keypoints(ex.handle(), 0);
    if (n < 0) throw std::runtime_error("AssignFeaturesToGrid before extract");
    std::vector<int32_t> start(COLS * ROWS + 1), items(n > 0 ? n : 1);
    if (jsorb_assign_features_to_grid(ex.handle(), 0, mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv, (int)COLS, (int)ROWS,
                                      start.data(), items.data()) != JSORB_OK)
        throw std::runtime_error(std::string("jsorb_assign_features_to_grid: ") + jsorb_last_error(ex.handle()));
    for (std::size_t i = 0; i < COLS; i++)
        for (std::size_t j = 0; j < ROWS; j++) {
            const std::size_t c = i * ROWS + j;
            mGrid[i][j].assign(items.begin() + start[c], items.begin() + start[c + 1]);
        }
}

} // namespace Jetson_SLAM

#endif // JSORB_COMPAT_HPP
