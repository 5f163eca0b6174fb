#!/bin/bash
# round 5: single-frame latency of the reference-shaped C++ driver (4 rotating pairs) under the knobs that change what a frame enqueues
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
python - <<PY
import numpy as np
from jetson_slam_amd.synth import synth_stereo_pair
ps = [synth_stereo_pair(s, 480, 752) for s in (1, 2, 3, 4)]
np.concatenate([p[0].ravel() for p in ps]).tofile('/tmp/fl.raw'); np.concatenate([p[1].ravel() for p in ps]).tofile('/tmp/fr.raw')
PY
run() { name=$1; shift; for i in 1 2; do env JSORB_JSON=1 JSORB_ROTATE_PAIRS=4 "$@" tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/fl.raw /tmp/fr.raw 400 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%-34s median %.1f us  p90 %.1f  extract %.1f  stereo %.1f  adopted %d' % (sys.argv[1], d['total_us_median'], d['total_us_p90'], d['extract_lr_us'], d['stereo_us'], d['speculative_matches_adopted']))" $name; done; }
run default            X=1
run frame_graph_0      JSORB_FRAME_GRAPH=0
run fused_0            JSORB_FUSED_DETECT_BLUR=0
run graph0_fused0      JSORB_FRAME_GRAPH=0 JSORB_FUSED_DETECT_BLUR=0
run persistent         JSORB_PERSISTENT_THREADS=1
run persistent_graph0  JSORB_PERSISTENT_THREADS=1 JSORB_FRAME_GRAPH=0
run spin0              JSORB_SPIN_WAIT=0
run kernel_upload_0    JSORB_KERNEL_UPLOAD=0
run fresh_syncedmem    JSORB_FRESH_SYNCEDMEM=1
run no_speculation     JSORB_SPECULATE=0
