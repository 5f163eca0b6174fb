#!/bin/bash
# One rocprofv3 --pmc pass with the given counters: tools/profile_pmc.sh <tag> "<COUNTER ...>" [extra bench.py arguments]
# -> gpurun_out/<tag>_pmc/ and a per-kernel summary on stdout
TAG=${1:-x}; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 6 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream $*"
rm -rf $O/${TAG}_pmc
rocprofv3 --pmc $CTRS --output-format csv -d $O/${TAG}_pmc -o p -- $B > $O/${TAG}_pmc.log 2>&1
python $ROOT/tools/pmc_summary.py $O/${TAG}_pmc 2>&1 | tail -12
