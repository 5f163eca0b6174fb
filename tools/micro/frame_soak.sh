#!/bin/bash
# Soak of the single-frame path with the speculative stereo match: 20 000 frames from two host threads, every frame compared with the first
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
for s, (h, w) in ((1, (480, 752)), (2, (240, 320))):
    l, r = synth_stereo_pair(s, h, w); l.tofile('/tmp/l%d.raw' % s); r.tofile('/tmp/r%d.raw' % s)
PY
JSORB_CHECK_EVERY_FRAME=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l1.raw /tmp/r1.raw 20000 2>&1 | tail -2
JSORB_CHECK_EVERY_FRAME=1 tools/micro/frame_latency 240 320 3 15 20 435.2 47.906 /tmp/l2.raw /tmp/r2.raw 20000 2>&1 | tail -2
for i in 1 2 3 4 5; do python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or api_sequence or two_host" 2>&1 | tail -1; done
