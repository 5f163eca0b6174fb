#!/usr/bin/env python3
"""Experiment: does splitting the 128-pair batch over G independent handle pairs (2G streams) raise throughput?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
H, W, L, tile, th, fx, bf = 480, 752, 8, 30, 20, 435.2, 47.906
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pairs = [synth_stereo_pair(1 + i, H, W) for i in range(16)]
left = torch.from_numpy(np.stack([pairs[i % 16][0] for i in range(P)])).cuda()
right = torch.from_numpy(np.stack([pairs[i % 16][1] for i in range(P)])).cuda()
for G in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else '1,2,4,8'.split(','))]:
    per = P // G
    hs = [(orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=per), orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, max_batch=per)) for _ in range(G)]
    if len(sys.argv) > 3 and sys.argv[3] == 'shared':
        keep = []
        for a, b in hs:
            st = torch.cuda.Stream(); keep.append(st)
            a.set_stream(st.cuda_stream); b.set_stream(st.cuda_stream)
    def step():
        for gi, (a, b) in enumerate(hs):
            a.extract_batch_device_async(left[gi * per:].data_ptr(), H * W, W, per)
            b.extract_batch_device_async(right[gi * per:].data_ptr(), H * W, W, per)
        for a, b in hs:
            orb.stereo_match_batch_async(a, b, bf / fx, bf)
    def fence():
        for a, b in hs: a.sync(); b.sync()
    for _ in range(5): step()
    fence()
    t0 = time.perf_counter(); n = 30
    for _ in range(n): step()
    fence()
    dt = time.perf_counter() - t0
    print("groups %d (%d pairs each, %d streams): %.0f pairs/s" % (G, per, 2 * G, n * P / dt))
    del hs
