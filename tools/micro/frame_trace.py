#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel + memory-copy trace of tools/micro/frame_latency: operations are grouped into frames by the gaps between
them, and for every operation slot of a frame the median start offset (from the frame's first operation) and duration are printed."""
import csv, glob, os, sys, statistics
d = sys.argv[1]
ops = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("jsorb::", "").replace("void ", "").split("<")[0], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Name", "?")), r.get("Stream_Id", "?")))
ops.sort()
frames, cur = [], []
for o in ops:
    if cur and o[0] - max(c[1] for c in cur) > 60000:      # > 60 us of GPU silence: next frame
        frames.append(cur); cur = []
    cur.append(o)
if cur:
    frames.append(cur)
sizes = [len(f) for f in frames]
mode = max(set(sizes), key=sizes.count)
good = [f for f in frames if len(f) == mode][5:]
print("frames: %d total, %d with the modal %d operations" % (len(frames), len(good), mode))
span = [max(o[1] for o in f) - f[0][0] for f in good]
print("GPU span of a frame: median %.1f us" % (statistics.median(span) / 1e3))
for i in range(mode):
    st = statistics.median(f[i][0] - f[0][0] for f in good) / 1e3
    du = statistics.median(f[i][1] - f[i][0] for f in good) / 1e3
    print("  %2d  +%7.1f us  %6.1f us  %-22s stream %s" % (i, st, du, good[0][i][2], good[0][i][3]))
