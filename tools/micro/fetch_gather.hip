// Calibration of rocprofv3's FETCH_SIZE for the access shapes of this path: kernels that read a KNOWN set of cache
// lines of a 2 GiB buffer (far beyond the 32 MiB of L2 and the 256 MiB Infinity Cache; every line at most once per kernel) with
//   stream16   64 lanes x 16 B consecutive (the staging loads of k_detect / k_blur / k_pyramid): every byte of every line used
//   line16     one 16-byte load per lane, every lane in its own 128-byte line (worst case of a gather)
//   half16     one 16-byte load per lane, every lane in its own 64-byte half line, both halves of a line touched by neighbouring lanes
//   patch48    three consecutive lanes read 48 contiguous bytes, the next three the row one pitch (2048 B) further (k_describe's patch rows)
//   desc32     two 16-byte loads per lane = one 32-byte descriptor per lane at a random 32-byte slot (k_stereo's candidate descriptors)
// Each kernel prints the bytes its lanes asked for, the distinct 64-byte and 128-byte blocks they lie in, and its duration; run under
//   rocprofv3 --pmc FETCH_SIZE -- tools/micro/fetch_gather      and compare FETCH_SIZE (KB) per kernel with those block counts.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_gather fetch_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned load16(const unsigned char *p) { const u32x4 v = *reinterpret_cast<const u32x4 *>(p); return v.x ^ v.y ^ v.z ^ v.w; }

// mode: 0 stream16, 1 line16, 2 half16, 3 patch48, 4 desc32
template <int MODE>
__global__ __launch_bounds__(256) void k_fetch(const unsigned char *buf, unsigned *out, unsigned long long n_threads)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_threads) return;
    unsigned acc = 0;
    if (MODE == 0) acc = load16(buf + t * 16);
    if (MODE == 1) acc = load16(buf + t * 128 + 16 * (t % 7));
    if (MODE == 2) acc = load16(buf + t * 64 + 16 * (t % 3));
    if (MODE == 3) {                                          // groups of 3 lanes: 48 contiguous bytes at an 8-byte aligned column of a 2048-byte row
        const unsigned long long g = t / 3, u = t % 3;
        const unsigned long long row = g;                     // every group its own row: lines are never shared between groups
        const unsigned col = (unsigned)((g * 40u) % 1984u) & ~7u;
        acc = load16(buf + row * 2048 + col + 16 * u);
    }
    if (MODE == 4) {                                          // a 32-byte slot per lane, slots shuffled inside blocks of 4096 slots
        const unsigned long long blk = t / 4096, i = t % 4096;
        const unsigned long long slot = blk * 4096 * 4 + ((i * 2731u) % 4096u) * 4 + (i % 4);      // one of 4 slots of a 128-byte line; every line visited once
        acc = load16(buf + slot * 32) ^ load16(buf + slot * 32 + 16);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const char *name, const unsigned char *buf, unsigned *out, unsigned long long n_threads, double asked, double b64, double b128)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned blocks = (unsigned)((n_threads + 255) / 256);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_fetch<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, n_threads);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-9s asked %8.1f MiB  in 64-byte blocks %8.1f MiB  in 128-byte lines %8.1f MiB  %7.3f ms  (%6.0f GB/s of 128-byte lines)\n", name, asked / 1048576.0,
           b64 / 1048576.0, b128 / 1048576.0, ms, b128 / (ms * 1e-3) / 1e9);
}

int main()
{
    const unsigned long long BYTES = 2ull << 30;
    unsigned char *buf; unsigned *out;
    if (hipMalloc(&buf, BYTES + 4096) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, BYTES + 4096);
    (void)hipDeviceSynchronize();
    const double G = (double)BYTES;
    run<0>("stream16", buf, out, BYTES / 16, G, G, G);
    run<1>("line16", buf, out, BYTES / 128, G / 8, G / 2, G);
    run<2>("half16", buf, out, BYTES / 64, G / 4, G, G);
    {   // patch48: one group of 3 lanes per 2048-byte row; a 48-byte run at an 8-aligned column touches 1 or 2 blocks - counted exactly
        const unsigned long long groups = BYTES / 2048;
        double b64 = 0, b128 = 0;
        for (unsigned long long g = 0; g < groups; g++) {
            const unsigned col = (unsigned)((g * 40u) % 1984u) & ~7u;
            b64 += 64.0 * ((col + 47) / 64 - col / 64 + 1);
            b128 += 128.0 * ((col + 47) / 128 - col / 128 + 1);
        }
        run<3>("patch48", buf, out, groups * 3, groups * 48.0, b64, b128);
    }
    run<4>("desc32", buf, out, BYTES / 128, G / 4, G / 2, G);
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}
