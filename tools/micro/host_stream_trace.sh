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
rows = list(csv.DictReader(open(glob.glob("$O/hs_trace/**/t_memory_copy_trace.csv", recursive=True)[0])))
c = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Bytes", r.get("Size", 0)) or 0)) for r in rows if "HOST_TO_DEVICE" in r["Direction"])
c = [x for x in c if x[2] > 1 << 20]
t0, t1 = c[len(c) // 3][0], c[-1][1]
sel = [x for x in c if x[0] >= t0]
busy = sum(e - s for s, e, _ in sel); tot = t1 - t0; byt = sum(b for _, _, b in sel)
print("big H2D copies: %d, span %.2f ms, sum of durations %.2f ms (%.0f %% of span; > 100 %% = concurrent engines), %.1f GB/s over the span, %.1f GB/s per copy while running"
      % (len(sel), tot / 1e6, busy / 1e6, 100.0 * busy / tot, byt / tot, byt / busy))
gaps = sorted((sel[i + 1][0] - max(x[1] for x in sel[:i + 1])) / 1e3 for i in range(len(sel) - 1))
print("gaps between consecutive copies (us): median %.1f, p90 %.1f, max %.1f" % (gaps[len(gaps) // 2], gaps[len(gaps) * 9 // 10], gaps[-1]))
for s, e, b in sel[:16]:
    print("  +%8.1f us  %7.1f us  %6.1f MB  %.1f GB/s" % ((s - t0) / 1e3, (e - s) / 1e3, b / 1e6, b / (e - s)))
PY
