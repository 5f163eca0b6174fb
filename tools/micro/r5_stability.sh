#!/bin/bash
# round 5: is anything flaky?  the whole GPU suite N times, the chain stress, the spill-arena test (concurrent chunk claims) over and over
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
N=${1:-6}
for i in $(seq $N); do python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
python tools/micro/chain_stress.py 600 f,a,g,i 2>&1 | tail -2
for i in $(seq 10); do python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "spill_arena or survivor_list_overflow" 2>&1 | tail -1; done
