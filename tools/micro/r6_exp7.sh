#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
X=$PWD/jetson_slam_amd/csrc/_build/variants/experiments/libjsorb.so
sed -i 's/--steps 12 --warmup 3 --min-time 0/--steps 30 --warmup 3 --min-time 0 $BARGS/' tools/micro/r6_timeline.sh
echo "== c2, alternating order"; tools/micro/r6_timeline.sh tl_c2_alt
echo "== c2, one order"; tools/micro/r6_timeline.sh tl_c2_one JSORB_LIBRARY=$X JSORB_LANE_ORDER=0
export BARGS="--config c3 --pairs 64"
echo "== c3, alternating order"; tools/micro/r6_timeline.sh tl_c3_alt
echo "== c3, one order"; tools/micro/r6_timeline.sh tl_c3_one JSORB_LIBRARY=$X JSORB_LANE_ORDER=0
