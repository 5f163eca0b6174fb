// Micro-benchmark: global_load_dwordx4 / dwordx2 throughput as a function of byte misalignment (L2-resident 8 MiB buffer,
// rows of 64 B per lane like the patch staging of k_describe).  Build: hipcc --offload-arch=gfx950 -O3 -o gmem_align gmem_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4), aligned(1)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2), aligned(1)));

template <int W>
__global__ __launch_bounds__(256) void k(const unsigned char *buf, unsigned *out, int iters, int stride, int off, unsigned mask)
{
    unsigned acc = 0;
    unsigned a = ((blockIdx.x * 256 + threadIdx.x) * (unsigned)stride) & mask;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned char *p = buf + ((a + u * 65536u) & mask) + off;
            if (W == 16) { u32x4 v = *reinterpret_cast<const u32x4 *>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            else { u32x2 v = *reinterpret_cast<const u32x2 *>(p); acc ^= v.x ^ v.y; }
        }
        a = (a + 4096u * 61u) & mask;
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <int W>
static void run(const unsigned char *buf, unsigned *out, int stride, int off, const char *what)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200, blocks = 256 * 16;
    const unsigned mask = (8u << 20) - 1u;
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, buf, out, 5, stride, off, mask);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, stride, off, mask);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks / 256 * 4 * iters * 4;       // wave-instructions per CU
    printf("%-36s width %2d stride %3d offset %2d : %7.3f ms  %6.1f clk/wave-instr/CU  %6.1f GB/s\n", what, W, stride, off, ms, ms * 1e-3 * 2.4e9 / winst,
           (double)blocks * 256 * iters * 4 * W / (ms * 1e-3) / 1e9);
}

int main()
{
    unsigned char *buf; unsigned *out;
    (void)hipMalloc(&buf, (8u << 20) + 4096); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 1, (8u << 20) + 4096);
    for (int off : {0, 1, 4, 8, 13}) run<16>(buf, out, 16, off, "dwordx4 consecutive lanes");
    for (int off : {0, 1, 4, 8, 13}) run<16>(buf, out, 64, off, "dwordx4 one per 64 B");
    for (int off : {0, 1, 3, 4}) run<8>(buf, out, 8, off, "dwordx2 consecutive lanes");
    for (int off : {0, 1, 3, 4}) run<8>(buf, out, 64, off, "dwordx2 one per 64 B");
    return 0;
}
