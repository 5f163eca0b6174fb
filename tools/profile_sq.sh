#!/bin/bash
# SQ instruction counters only (one rocprofv3 --pmc pass): gpurun_out/<tag>_sq_counters.csv
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 10 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream"
rm -rf $O/${TAG}_sq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/${TAG}_sq -o p -- $B > $O/${TAG}_sq.log 2>&1
python $ROOT/tools/pmc_summary.py $O/${TAG}_sq 2>&1 | tail -20
