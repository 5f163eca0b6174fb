#!/bin/bash
# round 6, experiment 1: the nominal-feature tiles (58 / 46 / 52) in k_detect's two forms, per-kernel times
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; cfg=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg 2>gpurun_out/r6_exp1_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp1_err.txt; }
for i in 1 2; do
run c2t58_fullplane "--config c2 --tile 58" X=1
run c2t58_compact   "--config c2 --tile 58" JSORB_DETECT_FULLPLANE=0
run c3t46_fullplane "--config c3 --tile 46 --pairs 64" X=1
run c3t46_compact   "--config c3 --tile 46 --pairs 64" JSORB_DETECT_FULLPLANE=0
run c5t52_fullplane "--config c5 --tile 52 --pairs 64" X=1
run c5t52_compact   "--config c5 --tile 52 --pairs 64" JSORB_DETECT_FULLPLANE=0
done
