#!/bin/bash
# round 5, experiment 1: what are more resident k_detect workgroups worth at the SAME band height?  (timing-only knock-out builds: wrong results)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=jetson_slam_amd/csrc/_build/variants
B="python bench.py --no-cpu-baseline --no-extras --min-time 1.5"
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-34s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 | python -c "$fmt" "$name"; }
for i in 1 2; do
run current               X=1
run ko16_natural          JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_NATURAL=1
run ko16_req21760         JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_REQUEST=21760
run ko16_req25600         JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_REQUEST=25600
run ko16_req33280         JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_REQUEST=33280
done
run ko16cap640_natural    JSORB_LIBRARY=$PWD/$V/ko16cap640/libjsorb.so JSORB_DETECT_LDS_NATURAL=1
run current_descpad       JSORB_DESCRIBE_LDS_PAD=7680
run ko16_natural_descpad  JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_NATURAL=1 JSORB_DESCRIBE_LDS_PAD=7680
run ko16_req21760_descpad JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_REQUEST=21760 JSORB_DESCRIBE_LDS_PAD=7680
EXTRA="--config c5 --pairs 64"
run c5_current            X=1
run c5_ko16_natural       JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_NATURAL=1
run c5_ko16_req21760      JSORB_LIBRARY=$PWD/$V/ko16/libjsorb.so JSORB_DETECT_LDS_REQUEST=21760
run c5_current_descpad    JSORB_DESCRIBE_LDS_PAD=7680
