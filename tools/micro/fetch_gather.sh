#!/bin/bash
# FETCH_SIZE calibration for gathers: tools/micro/fetch_gather under rocprofv3 --pmc (two separate passes), per-kernel rows printed raw.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/fg_a $O/fg_b
timeout 120 $ROOT/tools/micro/fetch_gather
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fg_a -o p -- $ROOT/tools/micro/fetch_gather > $O/fg_a.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/fg_b -o p -- $ROOT/tools/micro/fetch_gather > $O/fg_b.log 2>&1
python - <<PY
import csv, glob
for d in ("fg_a", "fg_b"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fetch" in r["Kernel_Name"]:
                print(r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"])
PY
