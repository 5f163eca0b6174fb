#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
X=$PWD/jetson_slam_amd/csrc/_build/variants/experiments/libjsorb.so
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-22s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 --profile-steps 0 2>gpurun_out/r6_exp3_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp3_err.txt; }
for i in 1 2; do
run default X=1
run expl_default JSORB_LIBRARY=$X
run blur_first_odd JSORB_LIBRARY=$X JSORB_LANE_ORDER=1
done
tools/micro/r6_timeline.sh r6_timeline
tools/micro/r6_timeline.sh r6_timeline_blurfirst JSORB_LIBRARY=$X JSORB_LANE_ORDER=1
