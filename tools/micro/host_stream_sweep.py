#!/usr/bin/env python3
"""Host-streamed regime (pinned host memory -> hipMemcpyAsync -> kernels) vs batch size, for the current lane settings (JSORB_MAX_LANES)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
H, W, L, tile, th, fx, bf = 480, 752, 8, 30, 20, 435.2, 47.906
pairs = [synth_stereo_pair(1 + i, H, W) for i in range(32)]
for P in ([int(os.environ['JSORB_SWEEP_P'])] if os.environ.get('JSORB_SWEEP_P') else (32, 64, 128, 256)):
    lh = torch.from_numpy(np.stack([pairs[i % 32][0] for i in range(P)])).pin_memory()
    rh = torch.from_numpy(np.stack([pairs[i % 32][1] for i in range(P)])).pin_memory()
    a = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
    b = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
    ln, rn = lh.numpy(), rh.numpy()
    def step():
        a.extract_batch_host_async(ln); b.extract_batch_host_async(rn); orb.stereo_match_batch_async(a, b, bf / fx, bf)
    for _ in range(4): step()
    a.sync(); b.sync()
    n = max(8, int(os.environ.get("JSORB_SWEEP_WORK", "16384")) // P); t0 = time.perf_counter()
    for _ in range(n): step()
    a.sync(); b.sync(); dt = time.perf_counter() - t0
    print("lanes<=%s  %3d pairs/batch: %.0f pairs/s, %.1f GB/s over PCIe" % (os.environ.get("JSORB_MAX_LANES", "4"), P, n * P / dt, n * P * 2 * H * W / dt / 1e9), flush=True)
    a.close(); b.close()
