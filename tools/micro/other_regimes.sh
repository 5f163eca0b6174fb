# the other bench.py configurations (the host-streamed and single-frame regimes are keys of the default bench.py line)
for c in c1 c3 c5; do python bench.py --config $c --pairs 64 --no-cpu-baseline --no-extras --min-time 1.0 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$c', d['value'], d['parity_vs_oracle'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['keypoints_image0'])"; done
python tools/micro/nms_ms_bench.py
