// jsorb_env.h - the environment variables libjsorb reads, in ONE place.
//
// Product switches (read in every build, documented in INTEGRATION.md "Runtime environment"):
//   JSORB_NO_ENV=1             the library does not touch the process environment (it sets GPU_MAX_HW_QUEUES=16 at load time otherwise)
//   JSORB_MAX_LANES=n          cap on the HIP streams a batch is split over (1 .. JSORB_MAX_LANES)
//   JSORB_LANE_MIN_MPX=x       level-0 megapixels a lane must carry before a batch is split (default 7)
//   JSORB_DETECT_FULLPLANE=0|1 force k_detect's compact / full-plane form on a batch handle
//   JSORB_SPECULATE=0|1        speculative stereo match behind a pair of single-frame extracts
//   JSORB_FRAME_GRAPH=0|1      single frames as one captured HIP graph
//   JSORB_THROUGHPUT_LAYOUT=1  a max_batch = 1 handle gets the batch launch layouts instead of the latency ones
// Experiment switches (launch layouts and fallback kernel paths forced by hand; A/B measurements and the variant tests): compiled in ONLY with
// -DJSORB_EXPERIMENTS - the `experiments` variant build of jetson_slam_amd/build.py.  The shipped library never reads them.
#pragma once
#include <cstdlib>

namespace jsorb {

inline const char *product_env(const char *name) { return getenv(name); }
#ifdef JSORB_EXPERIMENTS
inline const char *experiment_env(const char *name) { return getenv(name); }
#else
inline const char *experiment_env(const char *) { return nullptr; }
#endif
inline bool env_is(const char *v, int value) { return v && atoi(v) == value; }

} // namespace jsorb
