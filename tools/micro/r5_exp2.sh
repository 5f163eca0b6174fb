#!/bin/bash
# round 5, experiment 2: the compact k_detect (default for batch handles) against the full-plane form (JSORB_DETECT_FULLPLANE=1 = the round-4 kernel), LDS request sweep
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp2_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp2_err.txt; }
for i in 1 2; do
run fullplane             JSORB_DETECT_FULLPLANE=1
run compact               X=1
done
run compact_req21760      JSORB_DETECT_LDS_REQUEST=21760
run compact_req23040      JSORB_DETECT_LDS_REQUEST=23040
run compact_req26880      JSORB_DETECT_LDS_REQUEST=26880
run compact_budget19200   JSORB_DETECT_BUDGET=19200
run compact_budget24000   JSORB_DETECT_BUDGET=24000
EXTRA="--config c5 --pairs 64"
run c5_fullplane          JSORB_DETECT_FULLPLANE=1
run c5_compact            X=1
EXTRA="--config c3 --pairs 64"
run c3_fullplane          JSORB_DETECT_FULLPLANE=1
run c3_compact            X=1
