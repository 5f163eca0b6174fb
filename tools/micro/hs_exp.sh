cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
for rep in 1 2 3; do for q in 4 8 16; do
  echo -n "queues=$q: "; GPU_MAX_HW_QUEUES=$q tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 600 2>&1 | tail -1 | sed 's/.*adopted/adopted/'
done; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['host_streamed']['value'], d['host_streamed']['pcie_gb_per_s'], d['frame_latency_us']['total_us_median'], d['c4_batch64']['value'])"
