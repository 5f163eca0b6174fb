set -e
cd $(dirname $0)/../..
python - <<'PY'
import numpy as np, sys
sys.path.insert(0,'.')
from jetson_slam_amd.synth import synth_stereo_pair
l,r=synth_stereo_pair(1,480,752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
g++ -O2 -std=c++17 -I include tools/micro/frame_latency.cpp -L jetson_slam_amd -ljsorb -lpthread -Wl,-rpath,$PWD/jetson_slam_amd -o /tmp/frame_latency
/tmp/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 300
