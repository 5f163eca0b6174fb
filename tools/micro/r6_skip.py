#!/usr/bin/env python3
"""Marginal cost of every kernel INSIDE the overlapped 4-lane pipeline (round 6): the C2 batch step timed with one kernel's launches left out
(experiments build, JSORB_SKIP_KERNELS - results are wrong by construction, only the clock is read).  If a kernel's marginal cost is close to its
stand-alone duration the pipeline behaves serially for it; if it is much smaller, the other lanes' kernels fill the time it leaves.
Usage (GPU box): python tools/micro/r6_skip.py [config=c2] [pairs=128]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from jetson_slam_amd import build as jb
os.environ["JSORB_LIBRARY"] = jb.build_variant("experiments", *jb.VARIANTS["experiments"])
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, W, L, tile, th, fx, bf = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
nu = 32
sets = []
for si in range(4):
    prs = [synth_stereo_pair(1 + si * nu + i, H, W) for i in range(nu)]
    idx = np.arange(P) % nu
    sets.append((torch.from_numpy(np.stack([p[0] for p in prs])[idx]).to(dev), torch.from_numpy(np.stack([p[1] for p in prs])[idx]).to(dev)))
a = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
b = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
k = [0]
def step():
    l, r = sets[k[0] % 4]; k[0] += 1
    a.extract_batch_device_async(l.data_ptr(), H * W, W, P, keep=l)
    b.extract_batch_device_async(r.data_ptr(), H * W, W, P, keep=r)
    orb.stereo_match_batch_async(a, b, bf / fx, bf)
def timed(n=40):
    a.sync(); b.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    a.sync(); b.sync()
    return (time.perf_counter() - t0) / n * 1e3
cases = [("pyramid", 1), ("detect", 2), ("compact", 4), ("blur", 8), ("describe", 16), ("stereo", 32), ("median", 64),
         ("pyramid+blur", 1 | 8), ("detect+describe", 2 | 16), ("extract side (only stereo+median left)", 1 | 2 | 4 | 8 | 16),
         ("everything but detect", 127 & ~2), ("everything but blur", 127 & ~8), ("everything but describe", 127 & ~16), ("everything but pyramid", 127 & ~1)]
for _ in range(8): step()
base = min(timed() for _ in range(3))
print("%-50s %.4f ms/step" % ("all kernels", base))
for rep in range(2):
    for nm, mask in cases:
        os.environ["JSORB_SKIP_KERNELS"] = str(mask)
        for _ in range(3): step()
        t = min(timed() for _ in range(3))
        os.environ["JSORB_SKIP_KERNELS"] = "0"
        for _ in range(3): step()
        print("without %-42s %.4f ms/step   marginal cost %.4f ms" % (nm, t, base - t))
