#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel mean of each counter per dispatch."""
import csv, sys, collections, glob, os
for path in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("jsorb::", "").replace("void ", "").split("<")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("==", f)
        for k, d in acc.items():
            if not k.startswith("k_"): continue
            print("  %-12s" % k, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())), " (n=%d)" % len(next(iter(d.values()))))
