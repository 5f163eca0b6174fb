cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<PY
from jetson_slam_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(1, 480, 752); l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
echo "frame latency (C++ driver; the library constructor sets the queue default):"
for i in 1 2; do tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 400 2>&1 | tail -1; done
echo "with GPU_MAX_HW_QUEUES=4:"
GPU_MAX_HW_QUEUES=4 tools/micro/frame_latency 480 752 8 30 20 435.2 47.906 /tmp/l.raw /tmp/r.raw 400 2>&1 | tail -1
python tools/micro/host_stream_sweep.py 2>&1 | tail -4
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_q8.json 2> gpurun_out/bench_q8.err; python -c "
import json
d=json.loads(open('gpurun_out/bench_q8.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_vs_oracle'], d['env']); print(d['host_streamed']); print(d['frame_latency_us']); print(d['c4_batch64'])"
