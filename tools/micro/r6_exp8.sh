#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-26s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp8_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp8_err.txt; }
python -m pytest tests/test_gpu_parity.py -q -x -k "extract_and_stereo_bit_exact or batch_api or nms_ms or candidate_search or api_sequence_fuzz or two_threads" 2>&1 | tail -3
for CFG in "--config c2" "--config c3 --pairs 64" "--config c5 --pairs 64" "--config c2 --tile 58"; do
for i in 1 2; do
run "base $CFG"              JSORB_LIBRARY=$V/base/libjsorb.so
run "fused $CFG"             X=1
run "alt(fused/Bfirst) $CFG" JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=1
run "nofuse $CFG"            JSORB_LIBRARY=$V/experiments/libjsorb.so JSORB_LANE_ORDER=2
done
done
