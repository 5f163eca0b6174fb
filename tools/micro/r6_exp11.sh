#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
ROOT=$PWD
V=$PWD/jetson_slam_amd/csrc/_build/variants
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-20s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]) + " ".join("%s=%.4f" % (a[2:], b) for a, b in k.items() if b))'
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $CFG 2>gpurun_out/r6_exp11_err.txt | tail -1 | python -c "$fmt" "$name" || tail -5 gpurun_out/r6_exp11_err.txt; }
python -m pytest tests -m gpu -q -x -k "blur or extract_and_stereo_bit_exact or ptx or parameter_variants or full_size" 2>&1 | tail -3
for CFG in "--config c2" "--config c3 --pairs 64" "--config c5 --pairs 64"; do
echo "== $CFG"
for i in 1 2 3; do
run blur_down    JSORB_LIBRARY=$V/blur_down/libjsorb.so
run boustro      X=1
done
done
cd /tmp && export TMPDIR=/tmp
for v in blur_down boustro; do
  L=""; [ $v = blur_down ] && L=$V/blur_down/libjsorb.so
  rm -rf $ROOT/gpurun_out/pmc_$v
  JSORB_LIBRARY=$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/pmc_$v -o p -- python $ROOT/bench.py --steps 6 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream > /dev/null 2>&1
  echo "== FETCH_SIZE $v"; python $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_$v | grep -E "k_blur|k_describe"
  rm -rf $ROOT/gpurun_out/pmc_$v
done
