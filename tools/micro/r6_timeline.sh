#!/bin/bash
# round 6: what actually overlaps in the 4-lane pipeline - a rocprofv3 kernel trace of the DEFAULT multi-stream run, reduced to an occupancy-of-the-timeline table
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out
TAG=${1:-r6_timeline}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/${TAG} -o t -- python $ROOT/bench.py --steps 12 --warmup 3 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 > $O/${TAG}.log 2>&1
python $ROOT/tools/micro/r6_timeline.py $O/${TAG} | tee $O/${TAG}_summary.txt
rm -rf $O/${TAG}      # the raw trace is large; the summary is what is kept
