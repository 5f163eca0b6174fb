#!/usr/bin/env python3
"""Per-kernel issue cost of the vector instructions: profiles/valu_mix.json.

profiles/r04_valu_rate.txt (tools/micro/valu_rate2) shows two classes of vector instructions on gfx950: plain 32-bit VOP1 / VOP2 arithmetic and
logic (v_add / sub / and / or / xor / not / mov / lshrrev / ashrrev / fma / fmac / mul_f32 / bitop3 ...) issues every 2 clocks per SIMD as long
as no source operand is an SGPR, everything else (VOP3 three-operand integer forms, packed 16-bit and packed f32, v_perm / v_alignbit, conversions,
min / max, compares, v_lshlrev, 24-bit multiplies, DPP / SDWA forms, v_readlane, v_mbcnt ...) every 4 clocks; v_rcp_f32 and friends 8.  The SQ
counters do not separate the classes (SQ_ACTIVE_INST_VALU advances one quad-cycle per instruction whatever it is), so the mix of a kernel is
taken from its assembly: every vector instruction of the kernel, weighted 10^(loop depth) - the loop nesting clang annotates in its assembly
output - is classified, and the weighted mean issue cost (clocks per wave-instruction) is what bench.py multiplies the measured SQ_INSTS_VALU of
the kernel with.  It is a STATIC estimate (a hot path inside a rarely taken branch is over-weighted, an exact-path loop that almost never runs
as well); the bench line therefore carries the all-4-clock figure next to it.

Usage: python tools/valu_mix.py            (recompiles every kernel file to assembly with the build's flags; ~1 min)"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jetson_slam_amd import build as b      # noqa: E402

FULL_RATE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32",
             "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_add_u16", "v_accvgpr_write_b32",
             "v_accvgpr_read_b32", "v_mov_b64"}
SLOW8 = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_div_scale_f32", "v_div_fmas_f32",
         "v_div_fixup_f32"}
KERNELS = {"k_pyramid.hip": ["k_pyramid"], "k_detect.hip": ["k_detect"], "k_blur.hip": ["k_blur", "k_blur_compact"], "k_describe.hip": ["k_describe"],
           "k_stereo.hip": ["k_stereo", "k_median"], "k_compact.hip": ["k_compact_flat"]}


# the instantiation bench.py's default configuration launches (no mask, compass LUT, 6-bit early rejects, compact form)
PREFERRED = {"k_detect": "8k_detectILb0ELb1ELb1ELb1E", "k_blur_compact": "14k_blur_compactILi16E"}      # (k_blur_compact<16>: images with <= 4096 tiles; a one-lane batch - the profile runs - takes the fused launch)


def issue_clocks(line):
    m = re.match(r"\s+(v_\w+)\s*(.*)", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2).split(";")[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if base in SLOW8:
        return 8.0
    if op.endswith("_sdwa") or op.endswith("_dpp") or "row_" in rest or "quad_perm" in rest:
        return 4.0
    if base in FULL_RATE:
        srcs = [a.strip() for a in rest.split(",")][1:]
        if any(re.match(r"^(s\d+|s\[|vcc|exec|m0)", a) for a in srcs):      # an SGPR source halves the rate (literals and inline constants do not)
            return 4.0
        return 2.0
    return 4.0


def kernel_mix(asm_text, names):
    out = {}
    lines = asm_text.split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for i0, sym in starts:
        name = next((n for n in names if "%d%s" % (len(n), n) in sym), None)      # Itanium mangling: <length><name> - k_detect must not match k_detect_blur
        if not name:
            continue
        depth, tot_w, tot_c, n_static, n2 = 0, 0.0, 0.0, 0, 0
        for l in lines[i0 + 1:]:
            if ".Lfunc_end" in l:
                break
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l):
                m = re.search(r"Depth=(\d+)", l)
                depth = int(m.group(1)) if m else 0
                continue
            c = issue_clocks(l)
            if c is None:
                continue
            w = 10.0 ** depth
            tot_w += w
            tot_c += w * c
            n_static += 1
            n2 += c == 2.0
        key = name
        preferred = PREFERRED.get(name)
        if preferred:                                     # the instantiation the benchmark runs
            if preferred not in sym:
                continue
        elif key in out and out[key]["static_instructions"] >= n_static:      # several instantiations: keep the largest
            continue
        out[key] = {"clk_per_valu_instr": round(tot_c / max(tot_w, 1e-9), 3), "static_instructions": n_static, "static_full_rate_share": round(n2 / max(n_static, 1), 3)}
    return out


def main():
    res = {}
    for f, names in KERNELS.items():
        src = os.path.join(b.CSRC, f)
        asm = tempfile.mktemp(suffix=".s")
        subprocess.run([b._hipcc()] + b.FLAGS + b.FILE_FLAGS.get(f, []) + ["-S", "--cuda-device-only", "-o", asm, src], check=True, capture_output=True)
        res.update(kernel_mix(open(asm).read(), names))
        os.remove(asm)
    out = {"_source": "tools/valu_mix.py: static, loop-depth-weighted (10^depth) classification of every vector instruction of the kernel's gfx950 assembly "
                      "into the issue classes measured in profiles/r04_valu_rate.txt (2 / 4 / 8 clocks per wave-instruction per SIMD)",
           "_csrc_sha256": b.csrc_sha256(), "kernels": res}
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        print("%-16s %5.2f clk per VALU instruction (static: %4d instructions, %2.0f %% full rate)" % (k, v["clk_per_valu_instr"], v["static_instructions"], 100 * v["static_full_rate_share"]))
    print("wrote", path)


if __name__ == "__main__":
    main()
