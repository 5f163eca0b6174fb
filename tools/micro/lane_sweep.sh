#!/bin/bash
# Lane-schedule sweep on the GPU box: throughput of ONE handle pair (library lanes) vs lanes / minimum lane size.
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
B="python bench.py --no-cpu-baseline --no-extras --profile-steps 0 --min-time 1.0"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['parity_vs_oracle'])"; }
for l in 1 2 3 4 6 8; do
  run "JSORB_MAX_LANES=$l JSORB_LANE_MIN_MPX=3.5"
done
run "JSORB_MAX_LANES=4 JSORB_LANE_MIN_MPX=7"
run "JSORB_MAX_LANES=4 JSORB_LANE_STAGGER=1"
run "GPU_MAX_HW_QUEUES=8 JSORB_MAX_LANES=8 JSORB_LANE_MIN_MPX=3.5"
echo -n "groups=4 lanes=1: "; JSORB_MAX_LANES=1 $B --groups 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
