cd ${GRAFT_REPO_ROOT}
fmt='import json,sys; d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]; print("%-40s %8.1f pairs/s  %.4f ms/step  parity=%s  " % (sys.argv[1], d["value"], d["ms_per_step"], d["parity_vs_oracle"]))'
for cfg in "--config c3 --pairs 64" "--config c3 --pairs 128"; do
for i in 1 2; do
  for L in 4 6 8; do
    JSORB_MAX_LANES=$L JSORB_LANE_MIN_MPX=2 python bench.py --no-cpu-baseline --no-extras --min-time 1.0 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "lanes $L $cfg"
  done
done
done
