#!/bin/bash
# A/B on ONE box: variant `prev` (tools/micro/mk_prev.sh) against the current build, alternating; args: rounds, then bench configs (quoted)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=${1:-3}; shift
[ $# -eq 0 ] && set -- "--config c2"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-26s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
P=$PWD/jetson_slam_amd/csrc/_build/variants/prev/libjsorb.so
for cfg in "$@"; do
for i in $(seq $N); do
  B="python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg"
  JSORB_LIBRARY=$P $B 2>/dev/null | tail -1 | python -c "$fmt" "prev ${cfg#--config }"
  $B 2>/dev/null | tail -1 | python -c "$fmt" "new  ${cfg#--config }"
done
done
