import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # the library's default for the HIP runtime; torch initialises HIP first here

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """the CPU oracle binding (test infrastructure)"""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def orb():
    """the product binding; only GPU tests may create extractors through it"""
    # torch (device buffers, streams and collectives of some tests) brings up its HIP context before the library does: the order
    # bench.py uses.  (A filtered run in which torch's first device call came late - after the library, host threads and child
    # processes - saw torch report "no ROCm-capable device" on the test box; the full-suite order never did.)
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from jetson_slam_amd import orb as _orb
    return _orb


@pytest.fixture
def experiments_lib(orb, monkeypatch):
    """The `experiments` variant build (every translation unit with -DJSORB_EXPERIMENTS, csrc/jsorb_env.h) as the library behind `orb` for the
    duration of one test: the switches that force launch layouts and fallback kernel paths by hand exist only there - the shipped library
    never reads them."""
    from jetson_slam_amd import build as jb
    lib = orb.load_library(jb.build_variant("experiments", *jb.VARIANTS["experiments"]))
    monkeypatch.setattr(orb, "_lib", lib)
    return lib


CONFIGS = {
    # name: dict(h, w, L, tile, th, fx, bf)  - SURVEY.md 8(d)
    "tiny": dict(h=120, w=160, L=4, tile=12, th=20, fx=200.0, bf=20.0),
    "c1": dict(h=240, w=320, L=3, tile=15, th=20, fx=435.2, bf=47.906),
    "c2": dict(h=480, w=752, L=8, tile=30, th=20, fx=435.2, bf=47.906),
    "c3": dict(h=376, w=1241, L=8, tile=25, th=60, fx=718.86, bf=386.14),
    "c5": dict(h=720, w=1280, L=8, tile=20, th=20, fx=435.2, bf=47.906),
}


@pytest.fixture(scope="session")
def configs():
    return CONFIGS


@pytest.fixture(params=["latency", "throughput"])
def layout(request, monkeypatch):
    """Launch layouts of a handle created with max_batch = 1 (the reference's call shape): by default many short workgroups (8-row pyramid strips
    and blur bands, one tile row per k_detect workgroup: jsorb_create, Geometry::latency); JSORB_THROUGHPUT_LAYOUT=1 gives such a handle the
    layouts of the batch handles (long strips, bands of tile rows).  Tests that take this fixture run once with each."""
    if request.param == "throughput":
        monkeypatch.setenv("JSORB_THROUGHPUT_LAYOUT", "1")
    else:
        monkeypatch.delenv("JSORB_THROUGHPUT_LAYOUT", raising=False)
    return request.param
