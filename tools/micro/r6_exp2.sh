#!/bin/bash
# round 6, experiment 2: what clock does the chip actually run the 4-lane pipeline at?  (the VALU-issue ceiling is priced at the nominal 2.4 GHz)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
O=gpurun_out
( python bench.py --no-cpu-baseline --no-extras --min-time 8 > $O/r6_exp2_bench.json 2>$O/r6_exp2_err.txt ) &
BP=$!
sleep 1
: > $O/r6_exp2_clocks.txt
while kill -0 $BP 2>/dev/null; do
  ( date +%s.%N; rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i -E "sclk|mclk|fclk|power|GPU use" ) | tr '\n' ' ' >> $O/r6_exp2_clocks.txt; echo >> $O/r6_exp2_clocks.txt
  sleep 0.4
done
tail -c 600 $O/r6_exp2_bench.json; echo
awk 'NF>3' $O/r6_exp2_clocks.txt | tail -40
