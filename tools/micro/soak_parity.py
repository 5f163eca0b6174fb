import sys, numpy as np
sys.path.insert(0, '.')
from jetson_slam_amd import orb
from oracle import pyoracle as po
from jetson_slam_amd.synth import synth_stereo_pair
cfgs = {"c2": (480, 752, 8, 30, 20, 435.2, 47.906), "c5": (720, 1280, 8, 20, 20, 458.0, 50.0), "c3": (376, 1241, 8, 25, 60, 718.86, 386.14)}
for name, (H, W, L, tile, th, fx, bf) in cfgs.items():
    for nms in (False, True):
        gl = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, apply_nms_ms=nms); gr = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, apply_nms_ms=nms)
        kw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th, apply_nms_ms=nms)
        ol, orr = po.OracleExtractor(**kw), po.OracleExtractor(**kw)
        for seed in (501, 502, 503):
            l, r = synth_stereo_pair(seed, H, W)
            if seed == 503: l = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8); r = np.roll(l, -7, axis=1)
            kl, dl = gl.extract(l); kr, dr = gr.extract(r); ol.extract(l); orr.extract(r)
            assert np.array_equal(kl, ol.keypoints()) and np.array_equal(dl, ol.descriptors()), (name, nms, seed)
            assert np.array_equal(kr, orr.keypoints()) and np.array_equal(dr, orr.descriptors()), (name, nms, seed)
            mb = np.float32(bf) / np.float32(fx)
            u, d, st = orb.compute_stereo_matches(gl, gr, mb, bf); ou, od, ost = po.stereo_match(ol, orr, mb, bf)
            assert np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(d.view(np.uint32), od.view(np.uint32)), (name, nms, seed)
            print(name, nms, seed, "N", ol.n, "matched", ost["n_final"], "ok", flush=True)
