#!/bin/bash
# A/B on ONE box: the current build against named variants (csrc/_build/variants/<name>), alternating; usage: r6_abv.sh rounds "cfg" variant [variant ...]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=$1; CFG=$2; shift 2
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-22s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
V=$PWD/jetson_slam_amd/csrc/_build/variants
for i in $(seq $N); do
  B="python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG"
  $B 2>/dev/null | tail -1 | python -c "$fmt" "current"
  for v in "$@"; do JSORB_LIBRARY=$V/$v/libjsorb.so $B 2>/dev/null | tail -1 | python -c "$fmt" "$v"; done
done
