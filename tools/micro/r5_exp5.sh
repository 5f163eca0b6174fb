#!/bin/bash
# round 5, experiment 5: why does the pipeline not follow the kernel?  compact k_detect without the redo launch, and with larger LDS requests (fewer resident workgroups)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp5_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp5_err.txt; }
for i in 1 2; do
run fullplane             JSORB_DETECT_FULLPLANE=1
run compact               X=1
run compact_noredo        JSORB_EXPERIMENT_NO_REDO=1
run compact_req25600      JSORB_DETECT_LDS_REQUEST=25600
run compact_req32000      JSORB_DETECT_LDS_REQUEST=32000
run compact_req33280      JSORB_DETECT_LDS_REQUEST=33280
run compact_noredo_33280  JSORB_EXPERIMENT_NO_REDO=1 JSORB_DETECT_LDS_REQUEST=33280
run compact_lanes3        JSORB_MAX_LANES=3
run compact_lanes2        JSORB_MAX_LANES=2
done
