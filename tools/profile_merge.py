#!/usr/bin/env python3
"""Merge the per-configuration outputs of tools/profile_round.sh (tags <tag>, <tag>c3, <tag>c5 under gpurun_out/) into the files that
are committed under profiles/ and read by bench.py: hbm_traffic.json (= <tag>_hbm_traffic.json), valu_counters.json, and copies of
the kernel-stat / SQ-counter summaries.  Usage: tools/profile_merge.py r03"""
import csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
cfgs = {"c2": tag, "c3": tag + "c3", "c5": tag + "c5"}
pairs = {"c2": 128, "c3": 64, "c5": 64}

traffic = {"_source": "profiles/%s_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE passes of `bench.py --single-stream` "
                      "(tools/profile_round.sh %s / %sc3 / %sc5, merged by tools/profile_merge.py): c2 128 images per extract-side launch and 128 pairs per "
                      "stereo-side launch, c3 / c5 64" % (tag, tag, tag, tag),
           "_units": "bytes per kernel launch", "_raw_kb": {}}
valu = {"_source": "profiles/%s_sq_counters.csv (_c3, _c5): rocprofv3 --pmc SQ_INSTS_VALU of `bench.py --single-stream` (tools/profile_round.sh %s / %sc3 / %sc5)" % (tag, tag, tag, tag),
        "_units": "VALU wave-instructions per kernel launch; _pairs_per_launch images per extract-side launch and pairs per stereo-side launch"}
for cfg, t in cfgs.items():
    h = json.load(open(os.path.join(O, t + "_hbm_traffic.json")))
    traffic.setdefault("_note", h.get("_note", "") + "; calibrated for gathers too (profiles/r03_fetch_calibration.txt: the L2 fetches whole 128-byte lines whatever part "
                       "of them is asked for, one request per line tallied at 64 bytes), so the figure is 128 bytes x lines fetched for every kernel")
    traffic[cfg] = h[cfg]
    traffic["_raw_kb"][cfg] = h["_raw_kb"]
    sq = {r["kernel"]: r for r in csv.DictReader(open(os.path.join(O, t + "_sq_counters.csv")))}
    ks = {r["kernel"]: float(r["avg_us"]) for r in csv.DictReader(open(os.path.join(O, t + "_kernel_stats.csv"))) if r["kernel"].startswith("k_")}
    valu[cfg] = {"_pairs_per_launch": pairs[cfg], "avg_us": ks,      # stand-alone launch durations of the same single-stream run shape (rocprofv3 --kernel-trace --stats)
                 "extract_side": {k: int(float(sq[k]["SQ_INSTS_VALU"])) for k in sorted(sq) if k not in ("k_stereo", "k_median")},
                 "stereo_side": {k: int(float(sq[k]["SQ_INSTS_VALU"])) for k in ("k_stereo", "k_median") if k in sq}}
    suffix = "" if cfg == "c2" else "_" + cfg
    shutil.copy(os.path.join(O, t + "_kernel_stats.csv"), os.path.join(P, "%s_kernel_stats%s.csv" % (tag, suffix)))
    shutil.copy(os.path.join(O, t + "_sq_counters.csv"), os.path.join(P, "%s_sq_counters%s.csv" % (tag, suffix)))
for name in ("hbm_traffic.json", tag + "_hbm_traffic.json"):
    json.dump(traffic, open(os.path.join(P, name), "w"), indent=1)
sys.path.insert(0, ROOT)
from jetson_slam_amd import build as jb      # noqa: E402
valu["_csrc_sha256"] = jb.csrc_sha256()        # bench.py reports these counters only for the kernel sources they were measured on
json.dump(valu, open(os.path.join(P, "valu_counters.json"), "w"), indent=1)
import glob
raw = glob.glob(os.path.join(O, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True)
if raw:
    shutil.copy(raw[0], os.path.join(P, tag + "_rocprofv3_kernel_stats_raw.csv"))
print(json.dumps({c: sum(valu[c]["extract_side"].values()) * 2 + sum(valu[c]["stereo_side"].values()) for c in cfgs}))
