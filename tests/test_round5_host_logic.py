"""Round 5, host side (no GPU): the launch plan of a handle as jsorb_plan_launch reports it - the LDS layout of the compact k_detect, its spill
arena, and the LDS request of k_pyramid for every resampler form the kernel can instantiate (round-4 review: a level that needs THREE 16-byte
loads per lane runs the four-load form, whose bottom tap slot lies beyond what three loads ask for)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRANULE = 1280                      # LDS is handed out in granules of 1280 bytes on gfx950 (profiles/r04_lds_census.txt)
CONFIGS = {"c1": (240, 320, 3, 15), "c2": (480, 752, 8, 30), "c3": (376, 1241, 8, 25), "c5": (720, 1280, 8, 20), "c2_tile58": (480, 752, 8, 58),
           "tile128": (600, 800, 4, 128), "tile5": (200, 300, 5, 5)}


@pytest.fixture(scope="module")
def orb():
    import __graft_entry__ as g
    g.build()
    from jetson_slam_amd import orb as o
    return o


def test_pyramid_lds_request_covers_the_instantiated_resampler(orb):
    """k_pyramid's lanes park their tap rows in two LDS slots of 64 lanes x 16 bytes x NS each, behind 1792 bytes of bookkeeping; NS is what the kernel
    INSTANTIATES (1, 2 or 4), not the number of loads the level needs (3 runs the four-load form)."""
    seen = set()
    for sf, L in ((1.2, 8), (1.25, 7), (1.5, 6), (2.0, 4), (2.6, 3), (3.05, 3), (3.4, 3), (4.0, 2)):
        for w in (160, 752, 1241):
            p = orb.plan_launch(120, w, sf, L, 12, 12, max_batch=4)
            ns = [lv["pyr_ns_dispatched"] for lv in p["per_level"]]
            for lv in p["per_level"]:
                assert lv["pyr_ns_dispatched"] in (1, 2, 4) and lv["pyr_ns_dispatched"] >= lv["pyr_ns16"]
                seen.add((lv["pyr_ns16"], lv["pyr_ns_dispatched"]))
            assert p["pyramid_lds"] >= 1792 + 2 * 64 * 16 * max(ns), (sf, L, w, p["pyramid_lds"], ns)
    assert (3, 4) in seen and (2, 2) in seen and (1, 1) in seen      # the reviewer's case (scale 3.05^2 = 9.3) is among them


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_compact_detect_layout(orb, name, monkeypatch):
    """The compact k_detect (what every batch handle runs since round 6): at most 18 LDS granules per workgroup
    (7 workgroups per CU) unless a single tile row needs more, a pool of positives of >= 10 % of the band's region pixels, bands of whole tile rows,
    and a spill chunk that holds the largest band region (every pixel a positive) - so that no input can overflow it.  Single-image handles keep the
    full-plane form."""
    h, w, L, tile = CONFIGS[name]
    monkeypatch.delenv("JSORB_DETECT_FULLPLANE", raising=False)
    single = orb.plan_launch(h, w, 1.2, L, tile, tile, max_batch=1)
    assert single["compact"] == 0 and single["spill_chunks"] == 0 and all(lv["det_R"] == 1 for lv in single["per_level"])      # latency layout: one tile row per workgroup
    assert orb.plan_launch(h, w, 1.2, L, tile, tile, max_batch=64)["compact"] == 1      # the default choice for batch handles, whatever the tile size (jsorb_api.hip: plan_detect)
    monkeypatch.setenv("JSORB_DETECT_FULLPLANE", "0")
    p = orb.plan_launch(h, w, 1.2, L, tile, tile, max_batch=64)
    assert p["compact"] == 1
    if all(lv["det_R"] > 1 or tile <= 30 for lv in p["per_level"]):
        assert p["detect_lds"] <= 18 * GRANULE, p["detect_lds"]
    assert p["detect_lds"] <= 64 * 1024
    assert p["spill_chunks"] == 8 * 320 and p["spill_chunks"] // 8 >= 32 * 8      # per XCD: more chunks than workgroups that can be resident (32 CUs x 32 wave slots / 4 waves)
    blocks = 0
    scale = np.float32(1.0)
    for i, lv in enumerate(p["per_level"]):
        if i:
            scale = np.float32(1.2) * scale
        th = tile if i == 0 else int(np.float32(tile) * (np.float32(1.0) / scale))
        lw = w if i == 0 else int(np.float32(w) * (np.float32(1.0) / scale))
        tw = th
        ntw = (lw - 1) // tw + 1
        region = (lv["det_R"] * th + 2) * (lv["k_tiles"] * tw + 2)
        assert lv["pool"] >= max(256, -(-region // 10)), (i, lv, region)
        assert p["spill_chunk_entries"] >= region
        assert lv["score_stride"] % 2 == 0 and lv["score_stride"] >= lv["k_tiles"] * tw + 2
        assert 1 <= lv["det_R"] <= 8 and lv["det_R"] * th + 2 <= 255 and lv["det_R"] * lv["k_tiles"] <= 128
        blocks += -(-lv["tile_rows"] // lv["det_R"]) * (-(-ntw // lv["k_tiles"]))
    assert blocks == p["detect_blocks"]


def test_fullplane_knob_and_budget_knob_are_honoured():
    """JSORB_DETECT_FULLPLANE=1 / 0 forces the full-plane / compact form on a batch handle; JSORB_DETECT_BUDGET (experiments build only) raises the LDS budget the bands are chosen for."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom jetson_slam_amd import orb\np = orb.plan_launch(480, 752, 1.2, 8, 30, 30, max_batch=64)\n"
            "print(p['compact'], p['detect_lds'], p['detect_blocks'])") % ROOT
    def run(env):
        if "JSORB_DETECT_BUDGET" in env:      # an experiment switch: only the `experiments` variant build reads it (csrc/jsorb_env.h)
            from jetson_slam_amd import build as jb
            env = dict(env, JSORB_LIBRARY=jb.build_variant("experiments", *jb.VARIANTS["experiments"]))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-800:]
        return [int(v) for v in r.stdout.split()]
    base = run({})
    full = run({"JSORB_DETECT_FULLPLANE": "1"})
    big = run({"JSORB_DETECT_BUDGET": "30000"})
    assert base[0] == 1 and full[0] == 0 and full[1] > base[1]          # the full-plane form keeps plane, tile and lists side by side
    assert big[0] == 1 and big[1] > base[1] and big[2] <= base[2]        # a larger budget: taller bands, fewer workgroups
