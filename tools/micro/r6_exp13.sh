#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-20s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp13_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp13_err.txt; }
JSORB_LIBRARY=$V/blur_sw4/libjsorb.so python -m pytest tests -m gpu -q -x -k "blur or extract_and_stereo_bit_exact or ptx_chain or parameter_variants or full_size or mask" 2>&1 | tail -3
for CFG in "--config c2" "--config c5 --pairs 64"; do
for i in 1 2 3; do
run new17      X=1
run det18      JSORB_LIBRARY=$V/det18/libjsorb.so
run blur_sw4   JSORB_LIBRARY=$V/blur_sw4/libjsorb.so
run blur_sw4w7 JSORB_LIBRARY=$V/blur_sw4w7/libjsorb.so
done
done
