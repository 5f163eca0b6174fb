python tools/latency_bench.py 2>&1 | tail -12
for c in c1 c3 c5; do python bench.py --config $c --pairs 64 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$c', d['value'], d['parity_vs_oracle'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['keypoints_image0'])"; done
