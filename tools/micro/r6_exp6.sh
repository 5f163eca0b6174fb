#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-26s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp6_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp6_err.txt; }
for i in 1 2; do
run new_pf2                X=1
run pf3                    JSORB_LIBRARY=$V/blur_prefetch3/libjsorb.so
run pf4                    JSORB_LIBRARY=$V/blur_prefetch4/libjsorb.so
run pf2_one_order          JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=0
done
CFG="--config c3 --pairs 64"
for i in 1 2; do
run c3_base                JSORB_LIBRARY=$V/base/libjsorb.so
run c3_new_pf2             X=1
run c3_one_order           JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=0
run c3_mid512              JSORB_LIBRARY=$V/compact_mid512/libjsorb.so
run c3_pf3                 JSORB_LIBRARY=$V/blur_prefetch3/libjsorb.so
done
CFG="--config c3 --pairs 128"
for i in 1 2; do
run c3p128_base            JSORB_LIBRARY=$V/base/libjsorb.so
run c3p128_new_pf2         X=1
run c3p128_one_order       JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=0
done
