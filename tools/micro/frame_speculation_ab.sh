cd ${GRAFT_REPO_ROOT:-/root/repo}

python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
for s in 0 1 0 1; do echo "JSORB_SPECULATE=$s"; JSORB_SPECULATE=$s JSORB_TRACE_HOST=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 400 2>&1 | tail -4; done
