"""Round-6 host logic that needs no GPU: bench.py's multi-GPU self-check, the other image statistics of synth.py, the environment-switch
classification of csrc/jsorb_env.h."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rank(i, block=10.0, ag=0.02, bound=True, dev=None, pci=None, numa=0):
    return {"median_block_ms": block, "ms_per_step": block / 10, "all_gather_ms_mean": ag, "all_gather_ms_max": 2 * ag, "device_index": i if dev is None else dev,
            "pci": pci or "0000:%02x:00.0" % (0x10 + i), "numa_bound": bound, "numa_node": numa, "pid": 1000 + i}


def test_multi_self_check_passes_a_healthy_eight_rank_record():
    import bench
    r = bench.multi_self_check([_rank(i, block=10.0 + 0.03 * i, numa=i // 4) for i in range(8)])
    assert r["ok"] and all(v is True for v in r["checks"].values()), r
    assert "within 5 %" in r["diagnosis"]


def test_multi_self_check_names_what_is_wrong():
    import bench
    ranks = [_rank(i) for i in range(4)]
    ranks[2]["median_block_ms"] = 11.0                      # a 10 % straggler ...
    ranks[2]["numa_bound"] = False                          # ... that is not bound
    r = bench.multi_self_check(ranks)
    assert not r["ok"] and r["checks"]["no_straggler_over_5pct"] is False and r["checks"]["numa_bound"] is False
    assert "rank 2" in r["diagnosis"] and "not NUMA-bound" in r["diagnosis"] and "+10.0 %" in r["diagnosis"]
    # two ranks on one device / one PCI address
    ranks = [_rank(i) for i in range(4)]
    ranks[3]["device_index"] = 0
    ranks[3]["pci"] = ranks[0]["pci"]
    r = bench.multi_self_check(ranks)
    assert not r["ok"] and r["checks"]["distinct_devices"] is False and r["checks"]["distinct_pci"] is False
    # an expensive collective
    ranks = [_rank(i, ag=0.25 if i == 1 else 0.02) for i in range(2)]
    r = bench.multi_self_check(ranks)
    assert not r["ok"] and r["checks"]["all_gather_under_100us"] is False and "rank 1" in r["diagnosis"]


def test_multi_self_check_under_the_single_device_test_hook():
    """the 2- and 8-rank launch tests put every rank on cuda:0 over gloo: device / PCI / collective checks are reported as skipped, not failed"""
    import bench
    r = bench.multi_self_check([_rank(i, dev=0, pci="0000:10:00.0") for i in range(2)], single_device=True, gloo=True)
    assert r["ok"] and isinstance(r["checks"]["distinct_devices"], str) and isinstance(r["checks"]["all_gather_under_100us"], str)


def test_input_families_are_deterministic_and_distinct():
    from jetson_slam_amd.synth import INPUT_FAMILIES, synth_family_pair, synth_stereo_pair
    import hashlib
    seen = set()
    for fam in INPUT_FAMILIES:
        l, r = synth_family_pair(fam, 5, 120, 160)
        l2, r2 = synth_family_pair(fam, 5, 120, 160)
        assert l.dtype == np.uint8 and l.shape == (120, 160) and np.array_equal(l, l2) and np.array_equal(r, r2)
        assert not np.array_equal(l, synth_family_pair(fam, 6, 120, 160)[0])
        seen.add(hashlib.sha256(l.tobytes()).hexdigest())
    assert len(seen) == len(INPUT_FAMILIES)
    # statistics the families are named after
    noise = synth_family_pair("noise", 1, 240, 320)[0]
    assert 60 < noise.std() < 90
    sp = synth_family_pair("saltpepper", 1, 240, 320)[0]
    assert 0.2 < np.mean((sp < 8) | (sp > 247)) < 0.3
    flat = synth_family_pair("lowtexture", 1, 240, 320)[0]
    assert np.mean(np.abs(flat.astype(int) - 120) <= 1) > 0.7
    # the default generator is unchanged by the refactoring into _pair_from_clean (the committed goldens depend on it)
    l, r = synth_stereo_pair(7, 240, 320)
    assert hashlib.sha256(l.tobytes() + r.tobytes()).hexdigest()[:16] == "60855d8c3b8475f6"


def test_every_getenv_of_the_product_goes_through_jsorb_env_h():
    """csrc/jsorb_env.h is the ONE place that reads the environment: product switches by name there, everything else only in -DJSORB_EXPERIMENTS builds"""
    csrc = os.path.join(ROOT, "jetson_slam_amd", "csrc")
    product, experiment = set(), set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(csrc, f)).read()
        if f != "jsorb_env.h":
            assert not re.search(r"\bgetenv\s*\(", text), f
        product |= set(re.findall(r'product_env\("(\w+)"\)', text))
        experiment |= set(re.findall(r'experiment_env\("(\w+)"\)', text))
    assert product == {"JSORB_NO_ENV", "JSORB_MAX_LANES", "JSORB_LANE_MIN_MPX", "JSORB_DETECT_FULLPLANE", "JSORB_SPECULATE", "JSORB_FRAME_GRAPH", "JSORB_THROUGHPUT_LAYOUT"}, product
    assert not (product & experiment)
    header = open(os.path.join(csrc, "jsorb_env.h")).read()
    for name in product:
        assert name in header, name                        # every product switch is documented where it is read
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in product:
        assert name in doc, name


def test_source_digest_ignores_comments_and_sees_code():
    """profiles/valu_counters.json / valu_mix.json are stamped with a digest of the kernel sources' CODE: editing a comment must not drop the
    profile-derived view from the bench line, editing an instruction must (jetson_slam_amd/build.py:_code_only)."""
    from jetson_slam_amd import build as b
    src = 'int f(int a) {   // adds one\n    /* block\n comment */ return a + 1;      // "quoted" comment\n\n    const char *s = "// not a comment";\n}\n'
    same = 'int f(int a) {\n    return a + 1;   // another text\n    const char *s = "// not a comment";   /* x */\n}\n'
    other = src.replace("a + 1", "a + 2")
    assert b._code_only(src) == b._code_only(same)
    assert b._code_only(src) != b._code_only(other)
    assert '"// not a comment"' in b._code_only(src) and "adds one" not in b._code_only(src)


def test_committed_profiles_belong_to_the_committed_kernels():
    """Not an error when they do not (bench.py then leaves the VALU / traffic view out of its line and says so) - but it should not be the state of a
    finished round, so the mismatch shows up here as a skip with its reason."""
    import json
    import pytest
    from jetson_slam_amd import build as b
    sha = b.csrc_sha256()
    stale = [name for name in ("valu_counters.json", "valu_mix.json") if json.load(open(os.path.join(ROOT, "profiles", name))).get("_csrc_sha256") != sha]
    if stale:
        pytest.skip("profiles/%s were measured on other kernel sources than the ones in the tree: re-run tools/micro/r6_final_profile.sh" % ", ".join(stale))
