#!/usr/bin/env python3
"""Condense the outputs of tools/profile_round.sh into the files that are committed under profiles/."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, "%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("jsorb::", "").replace("void ", "").split("<")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items() if k.startswith("k_")}


stats = []
for f in glob.glob(os.path.join(O, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0].replace("jsorb::", "").replace("void ", "").split("<")[0]
        stats.append((name, int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"])))
stats.sort(key=lambda t: -t[1] * t[2])
with open(os.path.join(O, tag + "_kernel_stats.csv"), "w") as f:
    f.write("kernel,calls,avg_us,percent\n")
    for n, c, a, p in stats:
        f.write("%s,%d,%.2f,%.2f\n" % (n, c, a / 1e3, p))

fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
bench = {}
try:
    bench = json.loads(open(os.path.join(O, tag + "_bench.json")).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench json:", e)
cfg = sys.argv[2] if len(sys.argv) > 2 else bench.get("config", {}).get("name", "c2")
traffic = {cfg: {}, "_units": "bytes per kernel launch", "_note":
           "FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads "
           "(MI355X_MICROARCH.md, HBM section), and every staging load here is 16 B/lane, so traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "_raw_kb": {}}
for k in sorted(set(fetch) | set(write)):
    fk, wk = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
    traffic[cfg][k] = int((2 * fk + wk) * 1024)
    traffic["_raw_kb"][k] = {"FETCH_SIZE": round(fk, 1), "WRITE_SIZE": round(wk, 1), "GRBM_GUI_ACTIVE": write.get(k, {}).get("GRBM_GUI_ACTIVE")}
json.dump(traffic, open(os.path.join(O, tag + "_hbm_traffic.json"), "w"), indent=1)
with open(os.path.join(O, tag + "_sq_counters.csv"), "w") as f:
    names = sorted({c for d in sq.values() for c in d})
    f.write("kernel," + ",".join(names) + "\n")
    for k, d in sorted(sq.items()):
        f.write(k + "," + ",".join("%.4g" % d.get(c, 0) for c in names) + "\n")
print(open(os.path.join(O, tag + "_kernel_stats.csv")).read())
print(json.dumps(traffic[cfg]))
print(json.dumps({k: bench.get(k) for k in ("value", "ms_per_step", "parity_vs_oracle")}), json.dumps(bench.get("roofline")), json.dumps(bench.get("cpu_baseline")))
