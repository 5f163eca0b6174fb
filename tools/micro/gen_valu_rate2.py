#!/usr/bin/env python3
"""Generates tools/micro/valu_rate2.hip: sustained ISSUE RATE of the vector / scalar / LDS instruction classes the jsorb kernels are
made of, in clocks per wave-instruction per SIMD.

Why a second micro-benchmark: the round-3 one (valu_rate.hip) went through the compiler and several of its modes were single dependent
chains per lane.  Here every timed loop body is hand-written assembly: 64 instructions of ONE opcode (or a fixed mix) over 16 independent
destination registers, no memory traffic, so that what limits it is the issue rate alone.  Every wave reads s_memtime (shader clock)
and s_memrealtime (100 MHz constant clock) before and after its loop; the host prints
  clk/instr/SIMD  = waves_per_SIMD x delta(s_memtime) / instructions of one wave      (shader clocks, independent of DVFS)
  MHz             = delta(s_memtime) / delta(s_memrealtime) x 100                      (the clock the chip sustained in that kernel)
  wall clk/instr  = the same rate from hipEvent wall time at the nominal 2.4 GHz       (what a roofline priced at 2.4 GHz sees)
Run:  python tools/micro/gen_valu_rate2.py && hipcc --offload-arch=gfx950 -O2 -o tools/micro/valu_rate2 tools/micro/valu_rate2.hip
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# (name, template) ; {d} = destination register index 0..15, sources are v16..v31 (four register banks), {e} = even destination pair
OPS = [
    ("v_fma_f32", "v_fma_f32 v{d}, v{a}, v{b}, v{d}"),
    ("v_pk_fma_f32", "v_pk_fma_f32 v[{e}:{e1}], v[{a2}:{a21}], v[{b2}:{b21}], v[{e}:{e1}]"),
    ("v_pk_add_f32", "v_pk_add_f32 v[{e}:{e1}], v[{a2}:{a21}], v[{e}:{e1}]"),
    ("v_add_f32", "v_add_f32 v{d}, v{a}, v{d}"),
    ("v_add_u32", "v_add_u32 v{d}, v{a}, v{d}"),
    ("v_and_b32", "v_and_b32 v{d}, v{a}, v{b}"),
    ("v_or_b32", "v_or_b32 v{d}, v{a}, v{b}"),
    ("v_lshlrev_b32", "v_lshlrev_b32 v{d}, 3, v{a}"),
    ("v_lshl_or_b32", "v_lshl_or_b32 v{d}, v{a}, 8, v{b}"),
    ("v_and_or_b32", "v_and_or_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_add3_u32", "v_add3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_bfe_u32", "v_bfe_u32 v{d}, v{a}, 4, 8"),
    ("v_perm_b32", "v_perm_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_alignbyte_b32", "v_alignbyte_b32 v{d}, v{a}, v{b}, 3"),
    ("v_pk_sub_i16", "v_pk_sub_i16 v{d}, v{a}, v{b}"),
    ("v_pk_min_i16", "v_pk_min_i16 v{d}, v{a}, v{b}"),
    ("v_pk_max_i16", "v_pk_max_i16 v{d}, v{a}, v{b}"),
    ("v_pk_add_u16", "v_pk_add_u16 v{d}, v{a}, v{b}"),
    ("v_sad_u16", "v_sad_u16 v{d}, v{a}, v{b}, v{c}"),
    ("v_sad_u8", "v_sad_u8 v{d}, v{a}, v{b}, v{c}"),
    ("v_dot4_u32_u8", "v_dot4_u32_u8 v{d}, v{a}, v{b}, v{c}"),
    ("v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 v{d}, v{a}"),
    ("v_cvt_f32_ubyte2", "v_cvt_f32_ubyte2 v{d}, v{a}"),
    ("v_cvt_pk_f32_fp8", "v_cvt_pk_f32_fp8 v[{e}:{e1}], v{a}"),
    ("v_bcnt_u32_b32", "v_bcnt_u32_b32 v{d}, v{a}, v{b}"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 v{d}, s70, v{b}"),
    ("v_cmp_gt_i16 (vcc)", "v_cmp_gt_i16 vcc, v{a}, v{b}"),
    ("v_cmp_gt_u32 (sgpr pair)", "v_cmp_gt_u32 s[{sp}:{sp1}], v{a}, v{b}"),
        ("v_min_u32", "v_min_u32 v{d}, v{a}, v{b}"),
    ("v_max3_u32", "v_max3_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_mul_lo_u32", "v_mul_lo_u32 v{d}, v{a}, v{b}"),
    ("v_mul_u32_u24", "v_mul_u32_u24 v{d}, v{a}, v{b}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 v{d}, v{a}, v{b}, v{c}"),
    ("v_mov_b32", "v_mov_b32 v{d}, v{a}"),
    ("v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp v{d}, v{a} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_readlane_b32", "v_readlane_b32 s{sp}, v{a}, 5"),
    ("v_rcp_f32", "v_rcp_f32 v{d}, v{a}"),
    ("v_sub_u32", "v_sub_u32 v{d}, v{a}, v{b}"),
    ("v_xor_b32", "v_xor_b32 v{d}, v{a}, v{b}"),
    ("v_not_b32", "v_not_b32 v{d}, v{a}"),
    ("v_lshrrev_b32", "v_lshrrev_b32 v{d}, 1, v{a}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 v{d}, 7, v{a}"),
    ("v_mul_f32", "v_mul_f32 v{d}, v{a}, v{b}"),
    ("v_sub_f32", "v_sub_f32 v{d}, v{a}, v{b}"),
    ("v_fmac_f32", "v_fmac_f32 v{d}, v{a}, v{b}"),
    ("v_max_f32", "v_max_f32 v{d}, v{a}, v{b}"),
    ("v_max_u32", "v_max_u32 v{d}, v{a}, v{b}"),
    ("v_max_i32", "v_max_i32 v{d}, v{a}, v{b}"),
    ("v_min_f32", "v_min_f32 v{d}, v{a}, v{b}"),
    ("v_bitop3_b32", "v_bitop3_b32 v{d}, v{a}, v{b}, v{c} bitop3:0xe0"),
    ("v_bfi_b32", "v_bfi_b32 v{d}, v{a}, v{b}, v{c}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 v{d}, v{a}, 1, v{b}"),
    ("v_add_lshl_u32", "v_add_lshl_u32 v{d}, v{a}, v{b}, 1"),
    ("v_xad_u32", "v_xad_u32 v{d}, v{a}, v{b}, v{c}"),
    ("v_mad_i32_i24", "v_mad_i32_i24 v{d}, v{a}, v{b}, v{c}"),
    ("v_add_co_u32 (vcc)", "v_add_co_u32 v{d}, vcc, v{a}, v{b}"),
    ("v_pk_mov_b32", "v_pk_mov_b32 v[{e}:{e1}], v[{a2}:{a21}], v[{b2}:{b21}] op_sel:[0,1]"),
    ("v_pk_mul_f32", "v_pk_mul_f32 v[{e}:{e1}], v[{a2}:{a21}], v[{b2}:{b21}]"),
    ("v_pk_min_u16", "v_pk_min_u16 v{d}, v{a}, v{b}"),
    ("v_pk_sub_u16 clamp", "v_pk_sub_u16 v{d}, v{a}, v{b} clamp"),
    ("v_pk_lshrrev_b16", "v_pk_lshrrev_b16 v{d}, 1, v{a} op_sel_hi:[0,1]"),
    ("v_cvt_u32_f32", "v_cvt_u32_f32 v{d}, v{a}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 v{d}, v{a}"),
    ("v_cvt_f32_i32", "v_cvt_f32_i32 v{d}, v{a}"),
    ("v_rndne_f32", "v_rndne_f32 v{d}, v{a}"),
    ("v_floor_f32", "v_floor_f32 v{d}, v{a}"),
    ("v_med3_i32", "v_med3_i32 v{d}, v{a}, v{b}, v{c}"),
    ("v_add_u16", "v_add_u16 v{d}, v{a}, v{b}"),
    ("v_add_u32 sdwa BYTE_1", "v_add_u32_sdwa v{d}, v{a}, v{b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"),
    ("v_mov_b32 sdwa BYTE_2", "v_mov_b32_sdwa v{d}, v{a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2"),
    ("v_add_u32 dpp row_shr:1", "v_add_u32_dpp v{d}, v{a}, v{b} row_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_b32 dpp wave_shr:1", "v_mov_b32_dpp v{d}, v{a} wave_shr:1 row_mask:0xf bank_mask:0xf"),
    ("v_mov_b32 dpp quad_perm", "v_mov_b32_dpp v{d}, v{a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32 (sgpr operand)", "v_add_u32 v{d}, s70, v{b}"),
    ("v_and_b32 (literal)", "v_and_b32 v{d}, 0x7f7f7f7f, v{b}"),
    ("v_cndmask_b32 e32 (vcc set before the loop)", "v_cndmask_b32 v{d}, v{a}, v{b}, vcc"),
    ("v_cndmask_b32 e64 (sgpr pair)", "v_cndmask_b32 v{d}, v{a}, v{b}, s[70:71]"),
    ("v_cmp_gt_i32 e32 (vcc)", "v_cmp_gt_i32 vcc, v{a}, v{b}"),
    ("v_cmp_lt_i32 sdwa sext BYTE_0 (vcc)", "v_cmp_lt_i32_sdwa vcc, sext(v{a}), v{b} src0_sel:BYTE_0 src1_sel:DWORD"),
    ("v_accvgpr_write_b32", "v_accvgpr_write_b32 a{d}, v{a}"),
    ("v_accvgpr_read_b32", "v_accvgpr_read_b32 v{d}, a{d}"),
    ("ds_bpermute_b32 (LDS pipe)", "ds_bpermute_b32 v{d}, v32, v{a}"),
    ("ds_write_b16 (LDS only)", "ds_write_b16 v32, v{a}"),
    ("ds_read_u8 (LDS only)", "ds_read_u8 v{d}, v32"),
    ("s_bcnt1_i32_b64 (SALU only)", "s_bcnt1_i32_b64 s{sp}, s[70:71]"),
    ("s_and_saveexec_b64 + restore (2 SALU)", "s_and_saveexec_b64 s[{sp}:{sp1}], s[70:71]\\n s_mov_b64 exec, s[{sp}:{sp1}]"),
    ("s_add_u32 (SALU only)", "s_add_u32 s{sp}, s{sp}, 3"),
    ("s_and_b64 (SALU only)", "s_and_b64 s[{sp}:{sp1}], s[{sp}:{sp1}], s[70:71]"),
    ("ds_read_b32 (LDS only)", "ds_read_b32 v{d}, v32"),
    ("ds_read_b128 (LDS only)", "ds_read_b128 v[{q}:{q3}], v33"),
]

# mixes: list of templates cycled through the 64 slots
MIXES = [
    ("mix 1 SALU : 1 VALU (s_add_u32 / v_and_b32)", ["v_and_b32 v{d}, v{a}, v{b}", "s_add_u32 s{sp}, s{sp}, 3"]),
    ("mix 1 SALU : 2 VALU", ["v_and_b32 v{d}, v{a}, v{b}", "v_or_b32 v{d}, v{a}, v{b}", "s_add_u32 s{sp}, s{sp}, 3"]),
    ("mix 1 ds_read_b32 : 3 VALU", ["v_and_b32 v{d}, v{a}, v{b}", "v_or_b32 v{d}, v{a}, v{b}", "v_perm_b32 v{d}, v{a}, v{b}, v{c}", "ds_read_b32 v{dl}, v32"]),
    ("mix detect early reject (perm, pk_min, pk_max, pk_sub, and, cmp)", ["v_perm_b32 v{d}, v{a}, v{b}, v{c}", "v_pk_min_i16 v{d}, v{a}, v{b}", "v_pk_max_i16 v{d}, v{a}, v{b}", "v_pk_sub_i16 v{d}, v{a}, v{b}", "v_and_b32 v{d}, v{a}, v{b}", "v_cmp_gt_i16 vcc, v{a}, v{b}"]),
    ("mix blur (cvt_ubyte, pk_fma, pk_fma)", ["v_cvt_f32_ubyte0 v{d}, v{a}", "v_pk_fma_f32 v[{e}:{e1}], v[{a2}:{a21}], v[{b2}:{b21}], v[{e}:{e1}]", "v_pk_fma_f32 v[{e}:{e1}], v[{b2}:{b21}], v[{a2}:{a21}], v[{e}:{e1}]"]),
]

# dependent chains (latency): destination feeds the next instruction
CHAINS = [
    ("chain v_fma_f32", "v_fma_f32 v0, v0, v16, v17"),
    ("chain v_and_b32", "v_and_b32 v0, v0, v16"),
    ("chain v_perm_b32", "v_perm_b32 v0, v0, v16, v17"),
    ("chain v_pk_sub_i16", "v_pk_sub_i16 v0, v0, v16"),
    ("chain v_pk_fma_f32", "v_pk_fma_f32 v[0:1], v[0:1], v[16:17], v[18:19]"),
]


def fmt(t, k):
    d = k % 16
    e = 2 * (k % 8)
    a = 16 + (k * 5 + 1) % 16
    b = 16 + (k * 3 + 2) % 16
    c = 16 + (k * 7 + 3) % 16
    a2 = 16 + 2 * ((k + 1) % 8)
    b2 = 16 + 2 * ((k + 3) % 8)
    sp = 52 + 2 * (k % 8)
    q = 4 * (k % 4)
    return t.format(d=d, e=e, e1=e + 1, a=a, b=b, c=c, a2=a2, a21=a2 + 1, b2=b2, b21=b2 + 1, sp=sp, sp1=sp + 1, q=q, q3=q + 3, dl=d)


def body(templates, n=64):
    uses_lds = any(t.startswith("ds_") for t in templates)
    lines = []
    for k in range(n):
        t = templates[k % len(templates)]
        lines.append(fmt(t, k))
        if uses_lds and k % 16 == 15:
            lines.append("s_waitcnt lgkmcnt(0)")
    if uses_lds:
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def main():
    kernels = []
    for name, t in OPS:
        kernels.append((name, body([t])))
    for name, ts in MIXES:
        kernels.append((name, body(ts)))
    for name, t in CHAINS:
        kernels.append((name, [t] * 64))
    out = []
    out.append("// GENERATED by tools/micro/gen_valu_rate2.py - do not edit.  See that file for what is measured.")
    out.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n#include <vector>\n#include <algorithm>\n")
    clob = ", ".join('"v%d"' % i for i in range(40)) + ", " + ", ".join('"s%d"' % i for i in range(50, 76)) + ', ' + ", ".join('"a%d"' % i for i in range(16)) + ', "vcc", "scc", "memory"'
    for i, (name, lines) in enumerate(kernels):
        asm = "\\n\"\n        \"".join(lines)
        out.append("""
__global__ __launch_bounds__(1024) void k%d(unsigned long long *out, int iters)
{
    __shared__ unsigned lds[4096];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1; lds[threadIdx.x + 2048] = 2; lds[threadIdx.x + 3072] = 3;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile(
        "v_mbcnt_lo_u32_b32 v32, -1, 0\\n v_mbcnt_hi_u32_b32 v32, -1, v32\\n v_lshlrev_b32 v32, 2, v32\\n v_lshlrev_b32 v33, 2, v32\\n"
        "s_mov_b64 s[70:71], -1\\n s_mov_b64 vcc, -1\\n"
        "v_mov_b32 v0, v32\\n v_mov_b32 v1, v32\\n v_mov_b32 v2, v32\\n v_mov_b32 v3, v32\\n v_mov_b32 v4, v32\\n v_mov_b32 v5, v32\\n v_mov_b32 v6, v32\\n v_mov_b32 v7, v32\\n"
        "v_mov_b32 v8, v32\\n v_mov_b32 v9, v32\\n v_mov_b32 v10, v32\\n v_mov_b32 v11, v32\\n v_mov_b32 v12, v32\\n v_mov_b32 v13, v32\\n v_mov_b32 v14, v32\\n v_mov_b32 v15, v32\\n"
        "v_mov_b32 v16, 1.0\\n v_mov_b32 v17, 0.5\\n v_mov_b32 v18, 1.0\\n v_mov_b32 v19, 0.5\\n v_mov_b32 v20, v32\\n v_mov_b32 v21, 2.0\\n v_mov_b32 v22, v32\\n v_mov_b32 v23, 1.0\\n"
        "v_mov_b32 v24, 0.5\\n v_mov_b32 v25, v32\\n v_mov_b32 v26, 1.0\\n v_mov_b32 v27, v32\\n v_mov_b32 v28, 0.5\\n v_mov_b32 v29, 1.0\\n v_mov_b32 v30, v32\\n v_mov_b32 v31, 0.5\\n"
        "s_mov_b32 s52, 0\\n s_mov_b32 s53, 0\\n s_mov_b32 s54, 0\\n s_mov_b32 s55, 0\\n s_mov_b32 s56, 0\\n s_mov_b32 s57, 0\\n s_mov_b32 s58, 0\\n s_mov_b32 s59, 0\\n"
        "s_mov_b32 s60, 0\\n s_mov_b32 s61, 0\\n s_mov_b32 s62, 0\\n s_mov_b32 s63, 0\\n s_mov_b32 s64, 0\\n s_mov_b32 s65, 0\\n s_mov_b32 s66, 0\\n s_mov_b32 s67, 0\\n"
        "s_mov_b32 s72, %%4\\n"
        "s_barrier\\n"
        "s_memtime %%0\\n s_memrealtime %%2\\n s_waitcnt lgkmcnt(0)\\n"
        "1:\\n"
        "%s\\n"
        "s_sub_u32 s72, s72, 1\\n s_cmp_lg_u32 s72, 0\\n s_cbranch_scc1 1b\\n"
        "s_memtime %%1\\n s_memrealtime %%3\\n s_waitcnt lgkmcnt(0)\\n"
        : "=s"(t0), "=s"(t1), "=s"(r0), "=s"(r1) : "s"(iters) : %s);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
    }
}
""" % (i, asm, clob))
    out.append("struct K { const char *name; void (*fn)(unsigned long long *, int); int n; };\nstatic const K KS[] = {")
    for i, (name, lines) in enumerate(kernels):
        n = sum(1 for l in lines if not l.startswith("s_waitcnt"))
        out.append('    {"%s", k%d, %d},' % (name, i, n))
    out.append("};\n")
    out.append(r"""
int main(int argc, char **argv)
{
    const int iters = 2000;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz; iters %d x 64-instruction bodies; every CU gets ONE workgroup of 4 x W waves (W = waves per SIMD)\n", prop.name, cus, prop.clockRate, iters);
    printf("# clk = shader clocks (s_memtime) per wave-instruction per SIMD, median over all waves; MHz = shader clock sustained (s_memtime / s_memrealtime x 100 MHz);\n");
    printf("# wall = the same rate from hipEvent wall time priced at 2.4 GHz\n");
    printf("%-66s %s\n", "instruction", "   W=1: clk  MHz wall |   W=2: clk  MHz wall |   W=4: clk  MHz wall |   W=8: clk  MHz wall");
    unsigned long long *out; (void)hipMalloc(&out, sizeof(unsigned long long) * 2 * 32 * 2 * cus);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (const K &k : KS) {
        printf("%-66s", k.name);
        for (int w : {1, 2, 4, 8}) {
            const int threads = w <= 4 ? 256 * w : 1024, blocks = w <= 4 ? cus : 2 * cus;
            const int nw = blocks * threads / 64;
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, 50);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(2 * nw);
            (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
            std::vector<double> clk(nw), mhz(nw);
            for (int i = 0; i < nw; i++) { clk[i] = (double)h[2 * i]; mhz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 100.0 : 0.0; }
            std::sort(clk.begin(), clk.end()); std::sort(mhz.begin(), mhz.end());
            const double ninstr = (double)iters * k.n;
            printf(" | %5.2f %5.0f %5.2f", clk[nw / 2] / (ninstr * w), mhz[nw / 2], ms * 1e-3 * 2.4e9 / (ninstr * w));
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
""")
    with open(os.path.join(HERE, "valu_rate2.hip"), "w") as f:
        f.write("\n".join(out))
    print("wrote valu_rate2.hip with %d kernels" % len(kernels))


if __name__ == "__main__":
    main()
