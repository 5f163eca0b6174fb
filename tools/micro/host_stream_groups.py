#!/usr/bin/env python3
"""Host-streamed batch throughput (pinned host memory -> H2D -> kernels) vs number of handle pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
H, W, L, tile, th, fx, bf = 480, 752, 8, 30, 20, 435.2, 47.906
P = 128
pairs = [synth_stereo_pair(1 + i, H, W) for i in range(16)]
lh = torch.from_numpy(np.stack([pairs[i % 16][0] for i in range(P)])).pin_memory()
rh = torch.from_numpy(np.stack([pairs[i % 16][1] for i in range(P)])).pin_memory()
for G in (1, 2, 4):
    per = P // G
    hs = [(orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=per), orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=per)) for _ in range(G)]
    def step():
        for gi, (a, b) in enumerate(hs):
            a.extract_batch_host_async(lh.numpy()[gi * per:(gi + 1) * per]); b.extract_batch_host_async(rh.numpy()[gi * per:(gi + 1) * per])
        for a, b in hs: orb.stereo_match_batch_async(a, b, bf / fx, bf)
    def fence():
        for a, b in hs: a.sync(); b.sync()
    for _ in range(4): step()
    fence(); t0 = time.perf_counter(); n = 20
    for _ in range(n): step()
    fence(); dt = time.perf_counter() - t0
    print("host-streamed, %d handle pairs x %d pairs: %.0f pairs/s, %.1f GB/s over PCIe" % (G, per, n * P / dt, n * P * 2 * H * W / dt / 1e9), flush=True)
    del hs
