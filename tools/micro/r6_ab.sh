#!/bin/bash
# round 6 A/B on ONE box: the library saved as variant `base` (csrc/_build/variants/base/libjsorb.so) against the current build, alternating, C2 / C3 / C5
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=${1:-2}
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-14s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
BASE=$PWD/jetson_slam_amd/csrc/_build/variants/base/libjsorb.so
for i in $(seq $N); do
  for cfg in "--config c2" "--config c3 --pairs 64" "--config c5 --pairs 64"; do
    B="python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg"
    JSORB_LIBRARY=$BASE $B 2>/dev/null | tail -1 | python -c "$fmt" "base ${cfg#--config }"
    $B 2>/dev/null | tail -1 | python -c "$fmt" "new  ${cfg#--config }"
  done
done
