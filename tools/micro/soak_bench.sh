#!/bin/bash
# Parity soak through bench.py's own check (every unique pair of a step against the oracle: counts + checksum of keypoints, descriptors,
# uRight, depth): other seeds than the default line, the three full-size configurations.  ~1300 stereo pairs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
fmt='import json,sys; d=json.loads(sys.stdin.readline()); print("%s seed_base %s: %d pairs checked, parity=%s, %.0f pairs/s" % (sys.argv[1], sys.argv[2], d["parity_pairs_checked"], d["parity_vs_oracle"], d["value"]))'
O=${SOAK_OFFSET:-0}
for sb in $((2001+O)) $((4001+O)) $((6001+O)); do
  python bench.py --config c2 --pairs 128 --seed-base $sb --no-cpu-baseline --no-extras --min-time 0.3 2>/dev/null | python -c "$fmt" c2 $sb
done
for sb in $((3001+O)) $((7001+O)); do
  python bench.py --config c3 --pairs 128 --seed-base $sb --no-cpu-baseline --no-extras --min-time 0.3 2>/dev/null | python -c "$fmt" c3 $sb
done
for sb in $((5001+O)) $((9001+O)); do
  python bench.py --config c5 --pairs 48 --seed-base $sb --no-cpu-baseline --no-extras --min-time 0.3 2>/dev/null | python -c "$fmt" c5 $sb
done
python bench.py --config c1 --pairs 128 --seed-base 8001 --no-cpu-baseline --no-extras --min-time 0.3 2>/dev/null | python -c "$fmt" c1 8001
python tools/micro/soak_parity.py 2>&1 | tail -3
