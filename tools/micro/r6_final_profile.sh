#!/bin/bash
# round 6: the profile set behind profiles/r06_* (tools/profile_round.sh for C2 / C3 / C5; merged on the build host by tools/profile_merge.py r06)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
PROFILE_NO_BENCH=1 bash tools/profile_round.sh r06 2>&1 | tail -3
PROFILE_ARGS="--config c3 --pairs 64" PROFILE_CONFIG=c3 bash tools/profile_round.sh r06c3 2>&1 | tail -2
PROFILE_ARGS="--config c5 --pairs 64" PROFILE_CONFIG=c5 bash tools/profile_round.sh r06c5 2>&1 | tail -2
bash tools/profile_lds.sh r06 2>&1 | tail -12
# keep what the merge needs, drop the raw traces (gpurun_out is capped at 64 MiB)
for t in r06 r06c3 r06c5; do rm -rf gpurun_out/${t}_fetch gpurun_out/${t}_write gpurun_out/${t}_sq; find gpurun_out/${t}_trace -name "*kernel_trace.csv" -delete 2>/dev/null; done
rm -rf gpurun_out/r06_lds
du -sh gpurun_out
