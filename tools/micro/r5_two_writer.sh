#!/bin/bash
# round 5, the round-4 review's item 5: the two-writer delivery of uRight / depth (k_stereo stores the values into the pinned host mirror, k_median only the
# cut - dropped in round 4 after one unexplained failure of the GPU suite) rebuilt behind -DSTEREO_TWO_WRITER and hammered: does it fail again?
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants/two_writer/libjsorb.so
python - <<PY
import numpy as np
from jetson_slam_amd.synth import synth_stereo_pair, synth_adversarial_pair
ps = [synth_stereo_pair(s, 480, 752) for s in (1, 2, 3, 4)] + [synth_adversarial_pair(9, 480, 752)]
np.concatenate([p[0].ravel() for p in ps]).tofile('/tmp/l1.raw'); np.concatenate([p[1].ravel() for p in ps]).tofile('/tmp/r1.raw')
ps = [synth_stereo_pair(s, 240, 320) for s in (5, 6, 7, 8)]
np.concatenate([p[0].ravel() for p in ps]).tofile('/tmp/l2.raw'); np.concatenate([p[1].ravel() for p in ps]).tofile('/tmp/r2.raw')
PY
N=${SOAK_FRAMES:-30000}
echo "== frame soak, two-writer build (LD_PRELOAD), $N frames each"
for rep in 1 2; do
LD_PRELOAD=$V JSORB_ROTATE_PAIRS=5 JSORB_CHECK_EVERY_FRAME=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l1.raw /tmp/r1.raw $N 2>&1 | tail -1 | cut -c1-230
LD_PRELOAD=$V JSORB_ROTATE_PAIRS=4 JSORB_CHECK_EVERY_FRAME=1 JSORB_PERSISTENT_THREADS=1 tools/micro/frame_latency 240 320 3 15 20 435.2 47.906 /tmp/l2.raw /tmp/r2.raw $N 2>&1 | tail -1 | cut -c1-230
LD_PRELOAD=$V JSORB_ROTATE_PAIRS=4 JSORB_CHECK_EVERY_FRAME=1 JSORB_SPECULATE=0 tools/micro/frame_latency 240 320 3 15 20 435.2 47.906 /tmp/l2.raw /tmp/r2.raw $N 2>&1 | tail -1 | cut -c1-230
done
echo "== GPU tests that read single-frame stereo results, two-writer build, 8 times"
for i in $(seq 8); do JSORB_LIBRARY=$V python -m pytest tests -m gpu -q -p no:cacheprovider -k "extract_and_stereo or golden or ptx_chain or speculative or two_host or api_sequence or stale or full_hd or fuzz" 2>&1 | tail -1; done
echo "== chain stress, two-writer build"
JSORB_LIBRARY=$V python tools/micro/chain_stress.py 600 f,a,g,i 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -12
