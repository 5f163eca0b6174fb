#!/bin/bash
# round 6: is anything flaky after the scheduling changes (256-thread / fused compaction, alternating lane order, shared spill arena, bands walking both ways)?
# the whole GPU suite N times, the chain stress, the spill-arena + hand-back tests over and over, bench parity over other seeds and every configuration
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=${1:-4}
for i in $(seq $N); do python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
python tools/micro/chain_stress.py 400 f,a,g,i 2>&1 | tail -2
for i in $(seq 6); do python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "spill_arena or survivor_list_overflow or handback or tall_tiles" 2>&1 | tail -1; done
echo "== parity over other seeds (bench.py --seed-base; every unique pair of every input set against the oracle)"
for s in 1001 2001 5001; do python bench.py --no-cpu-baseline --no-extras --min-time 0.3 --profile-steps 0 --seed-base $s 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c2 seed-base', sys.argv[1], 'pairs checked', d['parity_pairs_checked'], 'parity', d['parity_vs_oracle'], 'pairs/s', d['value'])" $s; done
for c in "c3 --pairs 64" "c5 --pairs 64" "c2 --tile 58" "c3 --tile 46 --pairs 64" "c5 --tile 52 --pairs 64" "c1 --pairs 96"; do python bench.py --config $c --no-cpu-baseline --no-extras --min-time 0.3 --profile-steps 0 --seed-base 3001 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'seed-base 3001 pairs checked', d['parity_pairs_checked'], 'parity', d['parity_vs_oracle'], 'pairs/s', d['value'])" "$c"; done
echo "== frame soak"; bash tools/micro/frame_soak.sh 2>&1 | tail -6
