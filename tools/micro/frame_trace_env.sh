#!/bin/bash
# tools/micro/frame_trace.sh for several environments: frame_trace_env.sh "NAME=VAR=1 VAR2=2" ...   (output: gpurun_out/frame_trace_<NAME>.txt)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
g++ -O2 -std=c++17 -I $ROOT/include $ROOT/tools/micro/frame_latency.cpp -L $ROOT/jetson_slam_amd -ljsorb -lpthread -Wl,-rpath,$ROOT/jetson_slam_amd -o /tmp/frame_latency || exit 1
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  rm -rf $O/frame_trace_$name
  env $envs rocprofv3 --kernel-trace --output-format csv -d $O/frame_trace_$name -o t -- /tmp/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 100 > $O/frame_trace_$name.log 2>&1
  echo "== $name ($envs)"; python $ROOT/tools/micro/frame_trace.py $O/frame_trace_$name | tee $O/frame_trace_$name.txt
  rm -rf $O/frame_trace_$name
done
