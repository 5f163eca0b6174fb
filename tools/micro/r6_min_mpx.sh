#!/bin/bash
# level-0 megapixels a lane must carry before a batch is split (JSORB_LANE_MIN_MPX, product switch; default 7): usage r6_min_mpx.sh rounds "cfg" ...
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("%-44s %8.1f pairs/s  %.4f ms/step  parity=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]))'
for cfg in "$@"; do for i in $(seq $N); do for v in 7 5 4 3; do
  JSORB_LANE_MIN_MPX=$v python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "min_mpx $v ${cfg#--config }"
done; done; done
