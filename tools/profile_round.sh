#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel-trace stats + HBM traffic counters for one round.
# Usage: tools/profile_round.sh r01   -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
# --pmc passes are separate runs and never combined with trace domains other than the kernel trace.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# PROFILE_ARGS="--config c3 --pairs 64" profiles another configuration (no bench line then); tag it e.g. r03c3.  PROFILE_NO_BENCH=1 skips the
# bench line for the default configuration too (profiles first, then tools/profile_merge.py, then `python bench.py` reads the fresh counters).
if [ -z "$PROFILE_ARGS" ] && [ -z "$PROFILE_NO_BENCH" ]; then python $ROOT/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; fi
B="python $ROOT/bench.py --steps 10 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream $PROFILE_ARGS"   # one stream: per-kernel durations are not inflated by left/right overlap
rm -rf $O/${TAG}_trace $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- $B > $O/${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -o p -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_write -o p -- $B > $O/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/${TAG}_sq -o p -- $B > $O/${TAG}_sq.log 2>&1
python $ROOT/tools/profile_summary.py $TAG ${PROFILE_CONFIG:-}
