#!/bin/bash
# device-resident headline, alternating GPU_MAX_HW_QUEUES unset / 8 on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("  %s device-resident %8.1f pairs/s  parity=%s" % (sys.argv[1], d["value"], d["parity_vs_oracle"]))'
for i in 1 2 3; do
  for q in "" 8; do
    export GPU_MAX_HW_QUEUES=$q; [ -z "$q" ] && unset GPU_MAX_HW_QUEUES
    python bench.py --no-cpu-baseline --no-extras --min-time 1.5 2>/dev/null | python -c "$fmt" "queues=${q:-default}"
  done
done
