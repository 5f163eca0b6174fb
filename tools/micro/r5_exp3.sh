#!/bin/bash
# round 5, experiment 3: variants of the compact k_detect (pipelined pool atomic, cooperative zeroing of the plane) against the full-plane form
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp3_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp3_err.txt; }
for i in 1 2; do
run fullplane             JSORB_DETECT_FULLPLANE=1
run compact               X=1
run compact_pipe          JSORB_LIBRARY=$V/cp_pipe/libjsorb.so
run compact_coop          JSORB_LIBRARY=$V/cp_coop/libjsorb.so
run compact_pipecoop      JSORB_LIBRARY=$V/cp_pipecoop/libjsorb.so
done
run compact_budget24320   JSORB_DETECT_BUDGET=24320
run compact_req23040      JSORB_DETECT_LDS_REQUEST=23040
EXTRA="--config c5 --pairs 64"
run c5_fullplane          JSORB_DETECT_FULLPLANE=1
run c5_compact            X=1
run c5_compact_pipecoop   JSORB_LIBRARY=$V/cp_pipecoop/libjsorb.so
EXTRA="--config c3 --pairs 64"
run c3_fullplane          JSORB_DETECT_FULLPLANE=1
run c3_compact            X=1
run c3_compact_pipecoop   JSORB_LIBRARY=$V/cp_pipecoop/libjsorb.so
