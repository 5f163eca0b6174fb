#!/usr/bin/env python3
"""Generate tests/golden/*.npz: small seeded stereo pairs + the oracle's outputs for them.

The reference has no tests, fixtures or CPU path and cannot be executed, so these vectors are
REGRESSION fixtures produced by this repository's oracle (oracle/jsorb_oracle.c), not outputs of the reference.
The float stages are pinned to the reference separately by tools/ptx_vectors.py (vectors interpreted from its PTX).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from jetson_slam_amd.synth import synth_stereo_pair
from oracle import pyoracle as po

CASES = {
    "tiny_160x120_L4_t12": dict(seed=3, h=120, w=160, L=4, tile=12, th=20, fx=200.0, bf=20.0),
    "c1_320x240_L3_t15": dict(seed=7, h=240, w=320, L=3, tile=15, th=20, fx=435.2, bf=47.906),
}
for name, c in CASES.items():
    l, r = synth_stereo_pair(c["seed"], c["h"], c["w"])
    kw = dict(height=c["h"], width=c["w"], n_levels=c["L"], tile_h=c["tile"], tile_w=c["tile"], th_fast_max=c["th"])
    ol, orr = po.OracleExtractor(**kw), po.OracleExtractor(**kw)
    ol.extract(l); orr.extract(r)
    u, d, st = po.stereo_match(ol, orr, c["bf"] / c["fx"], c["bf"])
    tx, ty, ts = ol.tiles()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"),
                        left=l, right=r, params=np.array([c["h"], c["w"], c["L"], c["tile"], c["th"]], np.int32),
                        calib=np.array([c["fx"], c["bf"]], np.float32),
                        kp_left=ol.keypoints(), desc_left=ol.descriptors(), kp_right=orr.keypoints(), desc_right=orr.descriptors(),
                        tile_x=tx, tile_y=ty, tile_score=ts, level1_left=ol.level_image(1), blur1_left=ol.level_blurred(1),
                        u_right=u, depth=d,
                        stats=np.array([st[k] for k in ("n_candidate_pairs", "n_corr_match", "n_depth", "n_final")], np.int32))
    print(name, ol.n, orr.n, st["n_final"], os.path.getsize(os.path.join(ROOT, "tests", "golden", name + ".npz")))
