#!/bin/bash
# round 5, experiment 11: lanes and step size with the compact k_detect
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5 --profile-steps 0"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("%-28s %8.1f pairs/s  %.4f ms/step  parity=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/r5_exp11_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r5_exp11_err.txt; }
run lanes4_p128           X=1
run lanes2_p128           JSORB_MAX_LANES=2
run lanes6_p128           JSORB_MAX_LANES=6 JSORB_LANE_MIN_MPX=4
run lanes8_p128           JSORB_MAX_LANES=8 JSORB_LANE_MIN_MPX=3
run lanes4_fullplane      JSORB_DETECT_FULLPLANE=1
EXTRA="--pairs 256 --input-sets 2"
run lanes4_p256           X=1
run lanes8_p256           JSORB_MAX_LANES=8
run lanes4_p256_fullplane JSORB_DETECT_FULLPLANE=1
EXTRA="--pairs 64"
run lanes4_p64            X=1
run lanes2_p64            JSORB_MAX_LANES=2
