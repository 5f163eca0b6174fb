#!/bin/bash
# round 5, experiment 10: k_blur - 32-row bands on the largest levels, aligned 8-byte stores (first strip starts 4 columns in front of the ROI)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp10_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp10_err.txt; }
for i in 1 2; do
run current               X=1
run blur_lead4            JSORB_LIBRARY=$V/blur_lead4/libjsorb.so
run blur_tall1            JSORB_LIBRARY=$V/blur_tall1/libjsorb.so
run blur_tall3            JSORB_LIBRARY=$V/blur_tall3/libjsorb.so
run blur_lead4_tall2      JSORB_LIBRARY=$V/blur_lead4_tall2/libjsorb.so
done
