#!/bin/bash
# round 5 soak: (1) 3 x 20 000 single frames rotating through different pairs against a non-speculating handle pair, (2) the reference-derived chains
# (C2, tiny, C3 + NMS-MS, adversarial) 300 times inside one process, (3) bench.py's all-pairs parity over other seeds
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
echo "== frame soak (tools/micro/frame_soak.sh)"; bash tools/micro/frame_soak.sh 2>&1 | tail -8
echo "== chain stress (tools/micro/chain_stress.py 300 f,a,g,i)"; python tools/micro/chain_stress.py 300 f,a,g,i 2>&1 | tail -4
echo "== parity over other seeds (bench.py --seed-base)"
for s in 1001 2001; do python bench.py --no-cpu-baseline --no-extras --min-time 0.3 --profile-steps 0 --seed-base $s 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c2 seed-base', sys.argv[1], 'pairs checked', d['parity_pairs_checked'], 'parity', d['parity_vs_oracle'], 'pairs/s', d['value'])" $s; done
for c in c3 c5; do python bench.py --config $c --pairs 64 --no-cpu-baseline --no-extras --min-time 0.3 --profile-steps 0 --seed-base 3001 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'seed-base 3001 pairs checked', d['parity_pairs_checked'], 'parity', d['parity_vs_oracle'], 'pairs/s', d['value'])" $c; done
