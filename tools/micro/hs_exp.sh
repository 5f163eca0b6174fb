cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
for i in 1 2; do
echo -n "threads per frame : "; tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 1000 2>&1 | tail -1
echo -n "persistent workers: "; JSORB_PERSISTENT_THREADS=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 1000 2>&1 | tail -1
echo -n "persistent, plain : "; JSORB_SPECULATE=0 JSORB_PERSISTENT_THREADS=1 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 1000 2>&1 | tail -1
done
