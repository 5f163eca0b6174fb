#!/bin/bash
# kernel-trace stats only (one rocprofv3 pass, single stream): gpurun_out/<tag>_trace ; prints the per-kernel averages
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 10 --warmup 2 --min-time 0 --no-cpu-baseline --no-extras --profile-steps 0 --single-stream $@"
rm -rf $O/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -o t -- $B > $O/${TAG}_trace.log 2>&1
python - "$O/${TAG}_trace" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "jsorb" in r["Name"]:
            print("%-40s calls %5s avg %9.1f us  total %6.2f %%" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
