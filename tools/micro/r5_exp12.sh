#!/bin/bash
# round 5, experiment 12: minimum lane size with the compact k_detect (fewer, larger lanes at small batches?)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.2 --profile-steps 0"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp12_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp12_err.txt; }
for mpx in 7 11.6 16; do
EXTRA="--pairs 64";                run c2_p64_mpx$mpx   JSORB_LANE_MIN_MPX=$mpx
EXTRA="--pairs 128";               run c2_p128_mpx$mpx  JSORB_LANE_MIN_MPX=$mpx
EXTRA="--config c3 --pairs 64";    run c3_p64_mpx$mpx   JSORB_LANE_MIN_MPX=$mpx
EXTRA="--config c5 --pairs 64";    run c5_p64_mpx$mpx   JSORB_LANE_MIN_MPX=$mpx
EXTRA="--tile 58";                 run c2t58_mpx$mpx    JSORB_LANE_MIN_MPX=$mpx
done
