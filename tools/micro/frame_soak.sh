#!/bin/bash
# Soak of the single-frame path with the speculative stereo match: 2 x 20 000 frames from two host threads that ROTATE through 5 / 4 different pairs
# (one of them the adversarial pair of PTX chain i); every frame is compared with what a second pair of handles, which never arms the speculative
# match, computed for that pair before the loop - a match that delivered the previous frame's result fails.  Round 5 (the round-4 soak fed one pair).
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
import numpy as np
from jetson_slam_amd.synth import synth_stereo_pair, synth_adversarial_pair
ps = [synth_stereo_pair(s, 480, 752) for s in (1, 2, 3, 4)] + [synth_adversarial_pair(9, 480, 752)]
np.concatenate([p[0].ravel() for p in ps]).tofile('/tmp/l1.raw'); np.concatenate([p[1].ravel() for p in ps]).tofile('/tmp/r1.raw')
ps = [synth_stereo_pair(s, 240, 320) for s in (5, 6, 7, 8)]
np.concatenate([p[0].ravel() for p in ps]).tofile('/tmp/l2.raw'); np.concatenate([p[1].ravel() for p in ps]).tofile('/tmp/r2.raw')
PY
N=${SOAK_FRAMES:-20000}
JSORB_ROTATE_PAIRS=5 JSORB_CHECK_EVERY_FRAME=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l1.raw /tmp/r1.raw $N 2>&1 | tail -2
JSORB_ROTATE_PAIRS=4 JSORB_CHECK_EVERY_FRAME=1 JSORB_FRESH_SYNCEDMEM=1 tools/micro/frame_latency 240 320 3 15 20 435.2 47.906 /tmp/l2.raw /tmp/r2.raw $N 2>&1 | tail -2
JSORB_ROTATE_PAIRS=4 JSORB_CHECK_EVERY_FRAME=1 JSORB_PERSISTENT_THREADS=1 tools/micro/frame_latency 240 320 3 15 20 435.2 47.906 /tmp/l2.raw /tmp/r2.raw $N 2>&1 | tail -2
for i in 1 2 3; do python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or api_sequence or two_host" 2>&1 | tail -1; done
