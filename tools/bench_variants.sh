for f in jetson_slam_amd/csrc/_build/variants/*.so; do echo $f; JSORB_LIBRARY=$PWD/$f python bench.py --profile-steps 5 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['parity_vs_oracle'], d['roofline']['kernel_ms_per_step'])"; done
