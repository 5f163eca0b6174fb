#!/bin/bash
# A/B of the single-frame latency on ONE box: tools/micro/frame_latency linked against a saved build (csrc/_build/variants/<name>) and against the
# current one, alternating.  Usage: tools/micro/frame_ab.sh [variant=head] [rounds=3] [frames=300]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=${1:-head}; N=${2:-3}; F=${3:-300}
python - <<'PY'
import sys
sys.path.insert(0, '.')
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
VD=$PWD/jetson_slam_amd/csrc/_build/variants/$V
g++ -O2 -std=c++17 -I include tools/micro/frame_latency.cpp -L $VD -ljsorb -lpthread -Wl,-rpath,$VD -o /tmp/frame_latency_$V || exit 1
g++ -O2 -std=c++17 -I include tools/micro/frame_latency.cpp -L jetson_slam_amd -ljsorb -lpthread -Wl,-rpath,$PWD/jetson_slam_amd -o /tmp/frame_latency_cur || exit 1
for i in $(seq $N); do
  for w in $V cur; do
    echo -n "$w: "; /tmp/frame_latency_$w 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw $F 2>&1 | tail -1
  done
done
