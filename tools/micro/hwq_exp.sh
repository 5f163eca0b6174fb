#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP multiplexes all streams over this many hardware queues, default 4) against the three regimes
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("  device-resident %8.1f pairs/s  parity=%s" % (d["value"], d["parity_vs_oracle"]))'
for q in "" 8; do
  echo "GPU_MAX_HW_QUEUES=$q"
  export GPU_MAX_HW_QUEUES=$q; [ -z "$q" ] && unset GPU_MAX_HW_QUEUES
  python bench.py --no-cpu-baseline --no-extras --min-time 1.5 2>/dev/null | python -c "$fmt"
  for l in 2 4; do echo "  host lanes $l:"; JSORB_HOST_LANES=$l python tools/micro/host_stream_sweep.py 2>&1 | tail -4 | sed 's/^/    /'; done
  tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 400 2>&1 | tail -1 | sed 's/^/  /'
done
