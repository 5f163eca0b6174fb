// Micro-benchmark: issue rate of the vector instructions the blur / pyramid kernels are made of (clk per wave-instruction
// per SIMD; 4.0 = full rate).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float wa, float wb, unsigned seed)
{
    f2 A[8];
    float q[16];
    unsigned u = seed + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; j++) A[j] = (f2){(float)j, (float)(j + threadIdx.x)};
#pragma unroll
    for (int j = 0; j < 16; j++) q[j] = (float)(threadIdx.x + j);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // v_pk_fma_f32, scalar weight (op_sel broadcast), VGPR pair operand
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) A[j] = __builtin_elementwise_fma((f2){wa, wa}, (f2){q[j], q[j + 8]}, A[j]);
        } else if (MODE == 1) {     // v_fma_f32 x2, scalar weight
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) { A[j].x = __builtin_fmaf(wa, q[j], A[j].x); A[j].y = __builtin_fmaf(wb, q[j + 8], A[j].y); }
        } else if (MODE == 2) {     // v_cvt_f32_ubyteN
#pragma unroll
            for (int r = 0; r < 8; r++) {
#pragma unroll
                for (int j = 0; j < 4; j++) { q[4 * (r & 3) + j] += (float)((u >> (8 * j)) & 0xFFu); }
                u = u * 1664525u + 1013904223u;
            }
        } else if (MODE == 3) {     // v_pk_fma_f32 all-VGPR operands
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) A[j] = __builtin_elementwise_fma((f2){q[(j + 1) & 15], q[(j + 9) & 15]}, (f2){q[j], q[j + 8]}, A[j]);
        } else if (MODE == 4) {     // v_pk_mul_f32
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) A[j] = A[j] * (f2){wa, wa};
        } else if (MODE == 5) {     // v_perm_b32 / integer
#pragma unroll
            for (int r = 0; r < 32; r++) u = __builtin_amdgcn_perm(u, seed, 0x0c010c00u + r) + u;
        } else if (MODE == 6) {     // v_pk_sub_i16
            typedef short s2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int r = 0; r < 32; r++) u = __builtin_bit_cast(unsigned, (s2)(__builtin_bit_cast(s2, u) - __builtin_bit_cast(s2, seed + r)));
        } else if (MODE == 7) {     // v_sad_u16
#pragma unroll
            for (int r = 0; r < 32; r++) u = __builtin_amdgcn_sad_u16(u, seed + r, u);
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += A[j].x + A[j].y;
#pragma unroll
    for (int j = 0; j < 16; j++) s += q[j];
    if (s == 1234.5f || u == 77u) out[0] = s;
}

template <int MODE>
static void run(const char *what, int instr_per_iter, int waves_per_simd)
{
    float *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * waves_per_simd;     // one 256-thread block = 1 wave per SIMD of a CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f, 0.5f, 3u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f, 3u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)waves_per_simd * iters * instr_per_iter;      // per SIMD
    printf("%-44s waves/SIMD %d : %7.3f ms  %5.2f clk/wave-instr/SIMD (2.4 GHz)\n", what, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / winst);
    (void)hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) { run<0>("v_pk_fma_f32 sgpr weight", 32, 1); run<1>("v_fma_f32 x2 sgpr weight", 64, 1); run<2>("v_cvt_f32_ubyte + add", 64 + 8 * 2, 1); run<3>("v_pk_fma_f32 vgpr operands", 32, 1); run<4>("v_pk_mul_f32", 32, 1); run<5>("v_perm_b32 + add", 64, 1); run<6>("v_pk_sub_i16", 32, 1); run<7>("v_sad_u16", 32, 1); }
        if (w == 2) { run<0>("v_pk_fma_f32 sgpr weight", 32, 2); run<3>("v_pk_fma_f32 vgpr operands", 32, 2); run<7>("v_sad_u16", 32, 2); }
        if (w == 4) { run<0>("v_pk_fma_f32 sgpr weight", 32, 4); run<1>("v_fma_f32 x2 sgpr weight", 64, 4); run<2>("v_cvt_f32_ubyte + add", 80, 4); run<3>("v_pk_fma_f32 vgpr operands", 32, 4); run<4>("v_pk_mul_f32", 32, 4); run<5>("v_perm_b32 + add", 64, 4); run<6>("v_pk_sub_i16", 32, 4); run<7>("v_sad_u16", 32, 4); }
        if (w == 8) { run<0>("v_pk_fma_f32 sgpr weight", 32, 8); run<3>("v_pk_fma_f32 vgpr operands", 32, 8); }
    }
    return 0;
}
