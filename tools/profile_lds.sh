#!/bin/bash
# LDS counters only (one rocprofv3 --pmc pass): bank conflicts and LDS busy cycles per kernel.  gpurun_out/<tag>_lds/
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 10 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream"
rm -rf $O/${TAG}_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_lds -o p -- $B > $O/${TAG}_lds.log 2>&1
python $ROOT/tools/pmc_summary.py $O/${TAG}_lds 2>&1 | tail -20
tail -3 $O/${TAG}_lds.log
