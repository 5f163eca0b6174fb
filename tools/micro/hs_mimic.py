#!/usr/bin/env python3
"""Why does bench.py's host_streamed leg measure less than host_stream_sweep.py?  Same measurement, optionally after a device-resident
batch on other (still alive) handles, as in bench.py."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
H, W, L, tile, th, fx, bf = 480, 752, 8, 30, 20, 435.2, 47.906
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
NU = 128 if "u128" in mode else 32
pairs = [synth_stereo_pair(1 + i, H, W) for i in range(NU)]
keep = []
if "dev" in mode:
    a = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=128)
    b = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=128)
    lefts = torch.from_numpy(np.stack([pairs[i % NU][0] for i in range(128)])).cuda(); rights = torch.from_numpy(np.stack([pairs[i % NU][1] for i in range(128)])).cuda()
    for _ in range(20):
        a.extract_batch_device_async(lefts.data_ptr(), H * W, W, 128, keep=lefts); b.extract_batch_device_async(rights.data_ptr(), H * W, W, 128, keep=rights)
        orb.stereo_match_batch_async(a, b, bf / fx, bf)
    a.sync(); b.sync()
    keep = [a, b, lefts, rights]
    if "closed" in mode:
        a.close(); b.close(); keep = []
P = 256
lh = torch.from_numpy(np.stack([pairs[i % NU][0] for i in range(P)])).pin_memory()
rh = torch.from_numpy(np.stack([pairs[i % NU][1] for i in range(P)])).pin_memory()
x = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
y = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=P)
ln, rn = lh.numpy(), rh.numpy()
if "closelate" in mode:
    keep[0].close(); keep[1].close(); keep = []
def step():
    x.extract_batch_host_async(ln); y.extract_batch_host_async(rn); orb.stereo_match_batch_async(x, y, bf / fx, bf)
for _ in range(4): step()
x.sync(); y.sync()
n = 200; t0 = time.perf_counter()
for _ in range(n): step()
x.sync(); y.sync(); dt = time.perf_counter() - t0
print("%-14s %.0f pairs/s, %.1f GB/s" % (mode, n * P / dt, n * P * 2 * H * W / dt / 1e9), flush=True)
