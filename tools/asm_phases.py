#!/usr/bin/env python3
"""Static instruction counts per source section of a HIP kernel: inserts `asm volatile("; MARK <n>")` before every source line
that contains a given marker comment (default: lines starting with '// ----' or containing '//@'), compiles to gfx950
assembly and counts VALU / SALU / LDS+VMEM instructions between the marks and inside each loop.
Usage: tools/asm_phases.py jetson_slam_amd/csrc/k_detect.hip [kernel-symbol-substring] [extra hipcc flags...]"""
import re, subprocess, sys, os, tempfile

src = sys.argv[1]
sym = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
lines = open(src).read().split("\n")
out, n = [], 0
for l in lines:
    st = l.strip()
    if (st.startswith("// ----") or "//@" in st) and l.startswith("    "):
        out.append('    asm volatile("; MARK S%02d %s");' % (n, re.sub(r'[^A-Za-z0-9 _:+-]', '', st)[:50]))
        n += 1
    out.append(l)
d = os.path.dirname(os.path.abspath(src))
tmp = os.path.join(d, "_asm_phases_tmp.hip")
open(tmp, "w").write("\n".join(out))
asm = tempfile.mktemp(suffix=".s")
try:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                    "-fno-fast-math", "-S", "--cuda-device-only", "-o", asm, tmp] + extra, check=True, capture_output=True)
finally:
    os.remove(tmp)
L = open(asm).read().split("\n")
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and sym in l]
for s0 in starts:
    end = [i for i, l in enumerate(L) if ".Lfunc_end" in l and i > s0][0]
    body = L[s0:end]
    print("==", L[s0].split(":")[0][:90])
    cur, cnt, order = "prologue", {}, ["prologue"]
    for l in body:
        m = re.search(r"; MARK (S\d+ .*)", l)
        if m:
            cur = m.group(1); order.append(cur); continue
        c = cnt.setdefault(cur, [0, 0, 0])
        if re.match(r"\s+v_", l): c[0] += 1
        elif re.match(r"\s+s_", l): c[1] += 1
        elif re.match(r"\s+(ds_|global_|buffer_|flat_|scratch_)", l): c[2] += 1
    for k in order:
        if k in cnt: print("  %-58s valu %4d salu %4d mem %3d" % (k, *cnt[k]))
    print("  total valu", sum(c[0] for c in cnt.values()))
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            seg = body[labels[m.group(1)]:i]
            v = sum(1 for x in seg if re.match(r"\s+v_", x))
            if v >= 8: print("    loop lines %d-%d: valu %d salu %d mem %d" % (labels[m.group(1)], i, v, sum(1 for x in seg if re.match(r"\s+s_", x)), sum(1 for x in seg if re.match(r"\s+(ds_|global_)", x))))
    vg = [l.strip() for l in L[end:end + 400] if "vgpr_count" in l or "vgpr_spill" in l or "lds_size" in l or "group_segment_fixed_size" in l]
    print("  ", vg[:4])
