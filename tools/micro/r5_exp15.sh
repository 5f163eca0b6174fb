#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp15_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp15_err.txt; }
for i in 1 2; do
run compact_23040         X=1
run compact_28160         JSORB_DETECT_BUDGET=28160
run compact_32000         JSORB_DETECT_BUDGET=32000
run fullplane             JSORB_DETECT_FULLPLANE=1
done
