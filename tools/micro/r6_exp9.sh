#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
X=$V/experiments/libjsorb.so
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-34s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp9_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp9_err.txt; }
for CFG in "--config c2 --tile 58" "--config c3 --tile 46 --pairs 64" "--config c5 --tile 52 --pairs 64"; do
for i in 1 2; do
run "base $CFG"                    JSORB_LIBRARY=$V/base/libjsorb.so
run "fullplane plain nofuse"       JSORB_LIBRARY=$X JSORB_LANE_ORDER=2
run "fullplane alt"                JSORB_LIBRARY=$X JSORB_LANE_ORDER=1
run "compact plain nofuse"         JSORB_LIBRARY=$X JSORB_LANE_ORDER=2 JSORB_DETECT_FULLPLANE=0
run "compact alt"                  JSORB_LIBRARY=$X JSORB_LANE_ORDER=1 JSORB_DETECT_FULLPLANE=0
run "compact fused"                JSORB_LIBRARY=$X JSORB_LANE_ORDER=0 JSORB_DETECT_FULLPLANE=0
done
done
