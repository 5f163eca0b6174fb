// Micro-benchmark: LDS read throughput of one CU-filling launch as a function of access width and misalignment.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_align lds_align.hip ; run on the GPU box.  Prints clk per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int WIDTH>
__device__ __forceinline__ unsigned lds_read(unsigned addr)
{
    if constexpr (WIDTH == 1) { unsigned v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr)); return v; }
    else if constexpr (WIDTH == 2) { unsigned v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr)); return v; }
    else if constexpr (WIDTH == 4) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); return v; }
    else if constexpr (WIDTH == 8) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); return (unsigned)v ^ (unsigned)(v >> 32); }
    else { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v.x ^ v.y ^ v.z ^ v.w; }
}

template <int WIDTH>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, int stride, int offset)
{
    __shared__ __align__(16) unsigned char lds[32768];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)lds;   // LDS addresses are 32-bit offsets
    unsigned a = (unsigned)((threadIdx.x * stride) & 16383) + offset;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        unsigned r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = lds_read<WIDTH>(a + u * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= r[u];
        a = (a + (acc & 0)) ;
    }
    if (acc == 0x12345678u) out[0] = acc;
    (void)base;
}

template <int WIDTH>
static void run(int stride, int offset, const char *what)
{
    unsigned *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;
    hipLaunchKernelGGL(k<WIDTH>, dim3(blocks), dim3(256), 0, 0, out, 10, stride, offset);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WIDTH>, dim3(blocks), dim3(256), 0, 0, out, iters, stride, offset);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per CU: blocks/256 CUs * 4 waves * iters * 8
    const double winst = (double)blocks / 256 * 4 * iters * 8;
    printf("%-44s width %2d stride %3d offset %d : %7.3f ms  %6.2f clk/wave-instr/CU (2.4 GHz)\n", what, WIDTH, stride, offset, ms, ms * 1e-3 * 2.4e9 / winst);
    hipFree(out);
}

int main()
{
    run<1>(1, 0, "u8 consecutive bytes");
    run<1>(4, 0, "u8 one per dword");
    run<1>(7, 0, "u8 stride 7");
    run<2>(2, 0, "u16 aligned consecutive");
    run<2>(2, 1, "u16 odd address consecutive");
    run<2>(4, 1, "u16 odd, one per dword");
    run<2>(4, 3, "u16 straddling dwords");
    run<2>(7, 0, "u16 stride 7 (mixed)");
    run<4>(4, 0, "b32 aligned");
    run<4>(4, 1, "b32 +1");
    run<4>(4, 2, "b32 +2");
    run<8>(8, 0, "b64 aligned");
    run<8>(8, 4, "b64 +4");
    run<8>(8, 1, "b64 +1");
    run<8>(16, 1, "b64 +1 stride 16");
    run<16>(16, 0, "b128 aligned");
    run<16>(16, 4, "b128 +4");
    run<16>(16, 8, "b128 +8");
    run<16>(16, 1, "b128 +1");
    return 0;
}
