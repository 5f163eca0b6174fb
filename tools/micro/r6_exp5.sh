#!/bin/bash
# round 6, experiment 5: blur prefetch depth, lanes once more with the small k_compact, C3 / C5 sanity
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-26s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp5_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp5_err.txt; }
for i in 1 2; do
run new                    X=1
run blur_prefetch2         JSORB_LIBRARY=$V/blur_prefetch2/libjsorb.so
run lanes2                 JSORB_MAX_LANES=2
run lanes3                 JSORB_MAX_LANES=3
run lanes6                 JSORB_MAX_LANES=6 JSORB_LANE_MIN_MPX=3
run lanes8                 JSORB_MAX_LANES=8 JSORB_LANE_MIN_MPX=3
done
CFG="--config c3 --pairs 64"
for i in 1 2; do
run c3_base        JSORB_LIBRARY=$V/base/libjsorb.so
run c3_new         X=1
run c3_prefetch2   JSORB_LIBRARY=$V/blur_prefetch2/libjsorb.so
done
CFG="--config c5 --pairs 64"
for i in 1 2; do
run c5_base        JSORB_LIBRARY=$V/base/libjsorb.so
run c5_new         X=1
run c5_prefetch2   JSORB_LIBRARY=$V/blur_prefetch2/libjsorb.so
done
