#!/usr/bin/env python3
"""Stage-by-stage HIP-vs-oracle comparison (debug aid; the pytest -m gpu suite is the gate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jetson_slam_amd.synth import synth_stereo_pair
from jetson_slam_amd import orb
from oracle import pyoracle as po

CONFIGS = {
    "c1": dict(h=240, w=320, L=3, tile=15, th=20, fx=435.2, bf=47.906),
    "c2": dict(h=480, w=752, L=8, tile=30, th=20, fx=435.2, bf=47.906),
    "c3": dict(h=376, w=1241, L=8, tile=25, th=60, fx=718.86, bf=386.14),
    "small": dict(h=120, w=160, L=4, tile=12, th=20, fx=200.0, bf=20.0),
}

def run(name, seed=1, verbose=True):
    c = CONFIGS[name]
    l, r = synth_stereo_pair(seed, c["h"], c["w"])
    okw = dict(height=c["h"], width=c["w"], n_levels=c["L"], th_fast_max=c["th"], tile_h=c["tile"], tile_w=c["tile"])
    ol, orr = po.OracleExtractor(**okw), po.OracleExtractor(**okw)
    ol.extract(l); orr.extract(r)
    gl = orb.ORBExtractor(c["h"], c["w"], 1.2, c["L"], 9, 14, 7, c["th"], None, c["tile"], c["tile"])
    gr = orb.ORBExtractor(c["h"], c["w"], 1.2, c["L"], 9, 14, 7, c["th"], None, c["tile"], c["tile"])
    bad = 0
    for tag, o, g, im in (("L", ol, gl, l), ("R", orr, gr, r)):
        kp, ds = g.extract(im)
        assert g.level_dims() == o.level_dims(), (g.level_dims(), o.level_dims())
        for lv in range(c["L"]):
            a, b = g.level_image(lv), o.level_image(lv)
            if not np.array_equal(a, b): bad += 1; print(name, tag, "level image", lv, "mismatch", (a != b).sum())
            a, b = g.level_image(lv, blurred=True), o.level_blurred(lv)
            if not np.array_equal(a, b): bad += 1; print(name, tag, "blur", lv, "mismatch", (a != b).sum(), np.argwhere(a != b)[:5])
        tx, ty, ts = g.tile_candidates(); ox, oy, os_ = o.tiles()
        for nm, a, b in (("tile_score", ts, os_), ("tile_x", tx, ox), ("tile_y", ty, oy)):
            if not np.array_equal(a, b):
                bad += 1; idx = np.nonzero(a != b)[0]
                print(name, tag, nm, "mismatch", len(idx), idx[:8], a[idx[:8]], b[idx[:8]])
        if g.n_keypoints() != o.n: bad += 1; print(name, tag, "N", g.n_keypoints(), o.n)
        else:
            okp, ods = o.keypoints(), o.descriptors()
            n = o.n
            for k, nm in enumerate(["x", "y", "score", "angle", "octave", "size"]):
                a, b = kp[k*n:(k+1)*n], okp[k*n:(k+1)*n]
                if not np.array_equal(a, b): bad += 1; print(name, tag, "kp", nm, "mismatch", (a != b).sum())
            if not np.array_equal(ds, ods): bad += 1; print(name, tag, "desc mismatch rows", (ds != ods).any(1).sum())
        if verbose: print(name, tag, "N =", g.n_keypoints(), "levels", g.level_n_keypoints())
    mb = c["bf"] / c["fx"]
    u, d, st = orb.compute_stereo_matches(gl, gr, mb, c["bf"])
    ou, od, ost = po.stereo_match(ol, orr, mb, c["bf"])
    if not (np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(d.view(np.uint32), od.view(np.uint32))):
        bad += 1; idx = np.nonzero(u.view(np.uint32) != ou.view(np.uint32))[0]
        print(name, "stereo mismatch", len(idx), idx[:8], u[idx[:8]], ou[idx[:8]])
    for k in ("n_candidate_pairs", "n_corr_match", "n_depth", "n_final"):
        if st[k] != ost[k]: bad += 1; print(name, "stereo stat", k, st[k], ost[k])
    if verbose: print(name, "stereo", st)
    print(name, "seed", seed, "PARITY OK" if bad == 0 else "PARITY FAIL (%d)" % bad, flush=True)
    return bad

if __name__ == "__main__":
    names = sys.argv[1:] or ["small", "c1", "c2"]
    tot = 0
    for n in names:
        tot += run(n)
    sys.exit(1 if tot else 0)
