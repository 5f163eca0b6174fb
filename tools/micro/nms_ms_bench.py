#!/usr/bin/env python3
"""Throughput with apply_nms_ms = 1 (KITTI04-12 / KAIST / realsense yamls), both modes, per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from jetson_slam_amd import orb
from jetson_slam_amd.synth import synth_stereo_pair
CFG = {"c2": (480, 752, 8, 30, 20, 435.2, 47.906), "c5": (720, 1280, 8, 20, 20, 458.0, 50.0)}
for name in sys.argv[1:] or ["c2", "c5"]:
    H, W, L, tile, th, fx, bf = CFG[name]
    P = 32
    pairs = [synth_stereo_pair(1 + i, H, W) for i in range(8)]
    left = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(P)])).cuda()
    right = torch.from_numpy(np.stack([pairs[i % 8][1] for i in range(P)])).cuda()
    for nms, gpu in [(False, True), (True, True), (True, False)]:
        mk = lambda: orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, apply_nms_ms=nms, nms_ms_mode_gpu=gpu, max_batch=P)
        G = 4
        hs = [(mk(), mk()) for _ in range(G)]
        def step():
            for a, b in hs:
                a.extract_batch_device_async(left.data_ptr(), H * W, W, P); b.extract_batch_device_async(right.data_ptr(), H * W, W, P)
            for a, b in hs: orb.stereo_match_batch_async(a, b, bf / fx, bf)
        def fence():
            for a, b in hs: a.sync(); b.sync()
        for _ in range(3): step()
        fence(); t0 = time.perf_counter(); n = 10
        for _ in range(n): step()
        fence(); dt = time.perf_counter() - t0
        a = hs[0][0]
        st = torch.cuda.Stream(); a.set_stream(st.cuda_stream); a.reset_kernel_timing(); a.enable_kernel_timing(True)
        for _ in range(3): a.extract_batch_device_async(left.data_ptr(), H * W, W, P)
        a.sync()
        kt = {k: round(v[0] / max(1, v[1]) * 1e3, 1) for k, v in a.kernel_times().items() if v[1]}
        print(name, "nms_ms=%s mode_gpu=%s: %.0f pairs/s, N0=%d, us per 32-image launch: %s" % (nms, gpu, n * P * G / dt, a.n_keypoints(0), kt), flush=True)
        del hs
