#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace csv of the multi-lane bench to: (1) how many jsorb kernels run at the same instant (time-weighted histogram),
(2) which PAIRS of kernel kinds overlap for how long, (3) per kind: sum of durations vs the union of its intervals, (4) idle time of the busy window."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "jsorb" not in n:
            continue
        kind = n.split("jsorb::")[1].split("<")[0].split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no jsorb kernels in the trace")
# keep the steady state: the longest stretch of launches without an idle gap of more than 0.3 ms (the timed block; the parity passes around it pause for the oracle),
# minus its first and last fifth
best, cur0, end = (0, 0), 0, rows[0][1]
for i in range(1, len(rows) + 1):
    if i == len(rows) or rows[i][0] - end > 300000:
        if i - cur0 > best[1] - best[0]:
            best = (cur0, i)
        cur0 = i
    if i < len(rows):
        end = max(end, rows[i][1])
n = best[1] - best[0]
rows = rows[best[0] + n // 5: best[1] - n // 5]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, k, q, st in rows:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
active = collections.Counter()
hist = collections.Counter(); pair = collections.Counter(); solo = collections.Counter()
last = ev[0][0]
for t, d, k in ev:
    dt = t - last
    if dt > 0:
        n = sum(active.values())
        hist[min(n, 9)] += dt
        kinds = sorted(x for x, c in active.items() if c > 0)
        if n == 1:
            solo[kinds[0]] += dt
        for i in range(len(kinds)):
            for j in range(i, len(kinds)):
                if i != j or active[kinds[i]] > 1:
                    pair[(kinds[i], kinds[j])] += dt
    active[k] += d
    last = t
tot = t1 - t0
print("steady-state window %.3f ms, %d launches, queues used: %d, streams: %d" % (tot / 1e6, len(rows), len({r[3] for r in rows}), len({r[4] for r in rows})))
print("kernels in flight (time share):  " + "  ".join("%d: %.1f %%" % (n, 100.0 * hist[n] / tot) for n in sorted(hist)))
dur = collections.Counter(); cnt = collections.Counter()
for s, e, k, q, st in rows:
    dur[k] += e - s; cnt[k] += 1
print("per kind: launches, mean duration in the overlapped run, share of the window (sum of durations / window), time it ran ALONE")
for k in sorted(dur, key=lambda k: -dur[k]):
    print("  %-18s %5d  %8.1f us  %6.1f %%   alone %5.1f %%" % (k, cnt[k], dur[k] / cnt[k] / 1e3, 100.0 * dur[k] / tot, 100.0 * solo[k] / tot))
print("pairs of kinds in flight together (share of the window; same kind twice = two launches of it):")
for (a, b), v in sorted(pair.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-16s + %-16s %5.1f %%" % (a, b, 100.0 * v / tot))

# the order of launches on one queue (first 30 of the window), to see what a lane looks like
q0 = rows[0][3]
seq = [(s_, e_, k_) for s_, e_, k_, q_, st_ in rows if q_ == q0][:30]
print("queue %s: " % q0 + " ".join("%s(%d)" % (k_.replace("k_", "")[:4], (e_ - s_) // 1000) for s_, e_, k_ in seq))
