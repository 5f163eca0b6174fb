#!/bin/bash
# A/B on ONE GPU box (clocks differ by up to 10 % between boxes): the current build against a saved build of the same ABI.
# Usage: tools/micro/ab.sh [variant-name=base] [rounds=2]   (variants live in jetson_slam_amd/csrc/_build/variants/<name>/libjsorb.so)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=${1:-base}; N=${2:-2}
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-8s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
for i in $(seq $N); do
  JSORB_LIBRARY=$PWD/jetson_slam_amd/csrc/_build/variants/$V/libjsorb.so $B 2>/dev/null | python -c "$fmt" $V
  $B 2>/dev/null | python -c "$fmt" current
done
