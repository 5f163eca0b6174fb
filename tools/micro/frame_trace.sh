#!/bin/bash
# GPU-side timeline of ONE reference-shaped frame (tools/micro/frame_latency.cpp): rocprofv3 kernel + memory-copy trace of 60 frames,
# then the median start offset / duration of every operation inside a frame.  Output: gpurun_out/frame_trace.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
rm -rf $O/frame_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/frame_trace -o t -- $ROOT/tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 60 > $O/frame_trace.log 2>&1
python $ROOT/tools/micro/frame_trace.py $O/frame_trace | tee $O/frame_trace.txt
