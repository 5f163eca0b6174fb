# runs bench.py once per alternative build under jetson_slam_amd/csrc/_build/variants/<name>/libjsorb.so (JSORB_LIBRARY selects a build of the same ABI)
for f in jetson_slam_amd/csrc/_build/variants/*/libjsorb.so; do echo $f; JSORB_LIBRARY=$PWD/$f python bench.py --no-cpu-baseline --no-extras --min-time 1.0 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['parity_vs_oracle'], d['roofline']['kernel_ms_per_step'])"; done
