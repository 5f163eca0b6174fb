#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-20s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp10_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp10_err.txt; }
for CFG in "--config c2" "--config c5 --pairs 64"; do
echo "== $CFG"
for i in 1 2; do
run new    X=1
for v in pyr6 pyr7 det16 det17 st6; do run $v JSORB_LIBRARY=$V/$v/libjsorb.so; done
done
done
