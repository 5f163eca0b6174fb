#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the bench.py contract):
 (1) single-pair latency through the reference-shaped synchronous calls (host images in, two host threads for L/R like
     Frame.cpp:107-110, then ComputeStereoMatches) - what a SLAM front-end sees per frame;
 (2) host-streamed batch throughput (pinned host memory -> hipMemcpy2DAsync -> kernels), the PCIe-inclusive rate."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair

H, W, L, tile, th, fx, bf = 480, 752, 8, 30, 20, 435.2, 47.906
mk = lambda B=1: orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=B)
l, r = synth_stereo_pair(1, H, W)
lp, rp = torch.from_numpy(l).pin_memory(), torch.from_numpy(r).pin_memory()
exl, exr = mk(), mk()


def frame(threads=True):
    if threads:
        t1 = threading.Thread(target=exl.extract, args=(lp.numpy(),)); t2 = threading.Thread(target=exr.extract, args=(rp.numpy(),))
        t1.start(); t2.start(); t1.join(); t2.join()
    else:
        exl.extract_batch_host_async(lp.numpy()[None]); exr.extract_batch_host_async(rp.numpy()[None])
        orb.stereo_match_batch_async(exl, exr, bf / fx, bf); exl.sync(); exr.sync(); return
    orb.compute_stereo_matches(exl, exr, bf / fx, bf)


for mode in (True, False):
    for _ in range(20):
        frame(mode)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        frame(mode)
    dt = (time.perf_counter() - t0) / n
    print("single pair, %s: %.1f us/pair = %.0f pairs/s" % ("reference-shaped sync calls + 2 host threads + result copies" if mode else "async enqueue of L, R, stereo + one sync", dt * 1e6, 1 / dt))

# host-streamed batch
P = 64
pairs = [synth_stereo_pair(1 + i, H, W) for i in range(16)]
lh = torch.from_numpy(np.stack([pairs[i % 16][0] for i in range(P)])).pin_memory()
rh = torch.from_numpy(np.stack([pairs[i % 16][1] for i in range(P)])).pin_memory()
bl, br = mk(P), mk(P)
def step():
    bl.extract_batch_host_async(lh.numpy()); br.extract_batch_host_async(rh.numpy()); orb.stereo_match_batch_async(bl, br, bf / fx, bf)
for _ in range(5): step()
bl.sync(); br.sync()
t0 = time.perf_counter(); n = 30
for _ in range(n): step()
bl.sync(); br.sync()
dt = time.perf_counter() - t0
print("host-streamed batch (pinned, %d pairs/step): %.0f pairs/s, %.2f GB/s over PCIe" % (P, n * P / dt, n * P * 2 * H * W / dt / 1e9))
