#!/bin/bash
# round 5: the profile set behind profiles/r05_* (tools/profile_round.sh for C2 / C3 / C5, LDS counters, full-plane k_detect for comparison)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
PROFILE_NO_BENCH=1 bash tools/profile_round.sh r05 2>&1 | tail -3
PROFILE_ARGS="--config c3 --pairs 64" PROFILE_CONFIG=c3 bash tools/profile_round.sh r05c3 2>&1 | tail -2
PROFILE_ARGS="--config c5 --pairs 64" PROFILE_CONFIG=c5 bash tools/profile_round.sh r05c5 2>&1 | tail -2
bash tools/profile_lds.sh r05 2>&1 | tail -12
JSORB_DETECT_FULLPLANE=1 bash tools/profile_trace.sh r05fp 2>&1 | tail -9
