#!/bin/bash
# lane orders on ONE box (experiments build, JSORB_LANE_ORDER): 1 = odd lanes k_blur first + even lanes' compaction inside the k_blur launch (the default for tiles <= 40),
# 0 = the fused k_blur_compact launch on every lane, 2 = plain order; usage: r6_lane_order.sh rounds "cfg" ["cfg" ...]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-34s %8.1f pairs/s  %.4f ms/step  parity=%s  T=%s " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"], __import__("re").search(r"cap (\d+)", d["config"]["workload"]).group(1)) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
X=$PWD/jetson_slam_amd/csrc/_build/variants/experiments/libjsorb.so
for cfg in "$@"; do
for i in $(seq $N); do
  for o in ${ORDERS:-1 0 2}; do
    JSORB_LIBRARY=$X JSORB_LANE_ORDER=$o python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "order $o ${cfg#--config }"
  done
done
done
