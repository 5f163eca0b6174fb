#!/bin/bash
# single frame: upload by the first kernel of the frame from a per-handle pinned buffer (default) against hipMemcpyAsync from pageable memory
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
for i in 1 2 3; do for k in 0 1; do
  echo -n "JSORB_KERNEL_UPLOAD=$k: "; JSORB_KERNEL_UPLOAD=$k JSORB_TRACE_HOST=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 1000 2>&1 | tail -3 | sed 's/.*extract x/extract x/' | tr '\n' ' '; echo
done; done
echo "persistent threads:"; for k in 0 1; do echo -n "JSORB_KERNEL_UPLOAD=$k: "; JSORB_PERSISTENT_THREADS=1 JSORB_KERNEL_UPLOAD=$k tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 1000 2>&1 | tail -1; done
