#!/usr/bin/env python3
"""What fits next to what on a CU: static resources of every kernel of the pipeline (from its gfx950 assembly: VGPRs, static LDS, workgroup size)
plus the dynamic LDS the launches request at a given configuration, and for every ordered pair (A, B) how many workgroups of B find room on a CU that
A fills.  The batch pipeline overlaps the kernels of four lanes; round 4 found that the LDS partition - 1280-byte granules, 128 per CU - decides which
kernel can start next to k_detect (profiles/r04_lds_counters.txt, "LDS request sweep").  This is the census for all pairs (static: it says what CAN
co-reside, not what the dispatcher does).

Usage: python tools/lds_census.py [detect_dynamic_lds=33280] [pyramid_dynamic_lds=3840]      (no GPU needed; ~1 min of hipcc)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jetson_slam_amd import build as b      # noqa: E402

GRAN, CU_LDS, SIMD_VGPR, MAX_WAVES_SIMD = 1280, 160 * 1024, 512, 8
KERNELS = {"k_pyramid.hip": "9k_pyramidILb0ELb0EE", "k_detect.hip": "8k_detectILb0ELb1ELb1ELb1EE", "k_compact.hip": "14k_compact_flatILi16ELi256EE", "k_blur.hip": "6k_blurE",
           "k_describe.hip": "10k_describeE", "k_stereo.hip": "8k_stereoE"}
MEDIAN = ("k_stereo.hip", "8k_medianILi32EE")


def metadata(path, flags):
    asm = tempfile.mktemp(suffix=".s")
    subprocess.run([b._hipcc()] + b.FLAGS + flags + ["-S", "--cuda-device-only", "-o", asm, path], check=True, capture_output=True)
    out, cur = {}, {}
    for line in open(asm):
        m = re.match(r"\s+\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", line)
        if m:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_count":          # last field of a kernel's record (alphabetical order in the metadata)
                out[cur["name"]] = dict(cur)
                cur = {}
    os.remove(asm)
    return out


def main():
    dyn = {"k_detect": int(sys.argv[1]) if len(sys.argv) > 1 else 23040, "k_pyramid": int(sys.argv[2]) if len(sys.argv) > 2 else 3840}
    rows = {}
    for f, key in list(KERNELS.items()) + [MEDIAN]:
        md = metadata(os.path.join(b.CSRC, f), b.FILE_FLAGS.get(f, []))
        name = next(n for n in md if key in n)
        short = re.sub(r"^\d+", "", key.split("I")[0].split("E")[0])
        k = md[name]
        vg = (int(k["vgpr_count"]) + 7) // 8 * 8
        lds = int(k["group_segment_fixed_size"]) + dyn.get(short, 0)
        wg_waves = int(k["max_flat_workgroup_size"]) // 64
        rows[short] = dict(vgpr=vg, lds=lds, gran=(lds + GRAN - 1) // GRAN, waves=wg_waves)
    print("kernel        VGPRs  LDS per workgroup (granules)  waves per workgroup   workgroups per CU alone (limit)")
    for n, r in rows.items():
        by_vgpr = min(MAX_WAVES_SIMD, SIMD_VGPR // r["vgpr"]) * 4 // r["waves"]
        by_lds = (CU_LDS // GRAN) // r["gran"] if r["gran"] else 10 ** 6
        r["alone"] = min(by_vgpr, by_lds)
        print("%-12s %5d  %7d B (%3d)                %2d                    %3d (%s)" % (n, r["vgpr"], r["lds"], r["gran"], r["waves"], r["alone"], "LDS" if by_lds < by_vgpr else "VGPRs / wave slots"))
    print("\nworkgroups of B (column) that still fit on a CU that A (row) fills with its own maximum:")
    names = list(rows)
    print("%-12s" % "A \\ B" + "".join("%12s" % n for n in names))
    for a in names:
        ra = rows[a]
        free_gran = CU_LDS // GRAN - ra["alone"] * ra["gran"]
        used_waves_simd = (ra["alone"] * ra["waves"] + 3) // 4
        free_vgpr = SIMD_VGPR - used_waves_simd * ra["vgpr"]
        line = "%-12s" % a
        for bn in names:
            rb = rows[bn]
            by_lds = free_gran // rb["gran"] if rb["gran"] else 10 ** 6
            waves_simd = min(MAX_WAVES_SIMD - used_waves_simd, free_vgpr // rb["vgpr"])
            by_reg = max(0, waves_simd) * 4 // rb["waves"]
            line += "%12d" % max(0, min(by_lds, by_reg))
        print(line)


if __name__ == "__main__":
    main()
