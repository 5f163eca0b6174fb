#!/bin/bash
# Copy-engine timeline of the host-streamed regime: rocprofv3 memory-copy + kernel trace of a few 256-pair batches, then the busy
# fraction of the H2D copies and the gaps between them.  Output: gpurun_out/host_stream_trace.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/hs_trace
JSORB_SWEEP_P=${1:-256} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/hs_trace -o t -- python $ROOT/tools/micro/host_stream_sweep.py > $O/hs_trace.log 2>&1
python - <<PY | tee $O/host_stream_trace.txt
import csv, glob
P = int("${1:-256}")
rows = list(csv.DictReader(open(glob.glob("$O/hs_trace/**/t_memory_copy_trace.csv", recursive=True)[0])))
c = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "HOST_TO_DEVICE" in r["Direction"])
c = [x for x in c if x[1] - x[0] > 50000]                      # the chunk uploads (the small ones are tables at handle creation)
sel = c[len(c) // 3:]                                          # steady state
t0, t1 = sel[0][0], sel[-1][1]
busy = sum(e - s for s, e in sel); tot = t1 - t0
per_step = 2 * P * 752 * 480                                   # bytes uploaded per step (left + right images)
chunks_per_step = 4 if P >= 40 else 2                          # two lanes per handle from 40 images on
steps = len(sel) / chunks_per_step
print("host-streamed regime, %d pairs per batch: %d chunk uploads in %.2f ms; copy engine busy %.0f %% of the span; %.1f GB/s over the span (%.1f GB/s while a copy runs) = %.0f pairs/s"
      % (P, len(sel), tot / 1e6, 100.0 * busy / tot, steps * per_step / tot, steps * per_step / busy, steps * P / (tot / 1e9)))
gaps = sorted((sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 1))
print("gaps between consecutive uploads (us): median %.1f, p90 %.1f, max %.1f" % (gaps[len(gaps) // 2], gaps[len(gaps) * 9 // 10], gaps[-1]))
k = list(csv.DictReader(open(glob.glob("$O/hs_trace/**/t_kernel_trace.csv", recursive=True)[0])))
kb = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in k if t0 <= int(r["Start_Timestamp"]) <= t1)
print("kernel durations in the same span, summed over the two lanes: %.2f ms (%.0f %% of the span)" % (kb / 1e6, 100.0 * kb / tot))
PY
