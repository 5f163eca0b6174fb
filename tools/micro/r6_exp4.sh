#!/bin/bash
# round 6, experiment 4: k_compact with small workgroups; LDS requests of k_detect that leave room for other kernels' workgroups; lane order
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-26s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp4_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp4_err.txt; }
python -m pytest tests/test_gpu_parity.py -q -x -k "extract_and_stereo_bit_exact or batch_api or nms_ms or candidate_search or frame_unpack" 2>&1 | tail -3
for i in 1 2; do
run base_compact1024        JSORB_LIBRARY=$V/base/libjsorb.so
run new_compact256          X=1
run compact512              JSORB_LIBRARY=$V/compact512/libjsorb.so
run c256+blurfirst          JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=1
run c256+req26624           JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_DETECT_LDS_REQUEST=26624
run c256+req26624+blurfirst JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_DETECT_LDS_REQUEST=26624 JSORB_LANE_ORDER=1
run c256+req30720           JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_DETECT_LDS_REQUEST=30720
done
CFG="--config c5 --pairs 64"
for i in 1 2; do
run c5_base        JSORB_LIBRARY=$V/base/libjsorb.so
run c5_new         X=1
run c5_compact512  JSORB_LIBRARY=$V/compact512/libjsorb.so
done
