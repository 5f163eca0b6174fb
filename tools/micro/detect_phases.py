#!/usr/bin/env python3
"""Where k_detect's waves spend their clocks: builds libjsorb with -DDET_TIMING (every wave adds the shader clocks of each phase to global
counters), runs batches of synthetic stereo images of one configuration through it and prints the split.
Usage (GPU box): python tools/micro/detect_phases.py [c2|c3|c5] [n_images]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jetson_slam_amd import build as b                              # noqa: E402

lib = b.build_variant("detect_timing", ["-DDET_TIMING"], ["k_detect.hip"])
os.environ["JSORB_LIBRARY"] = lib
import numpy as np                                                   # noqa: E402
import torch                                                         # noqa: E402
from jetson_slam_amd import orb                                      # noqa: E402
from jetson_slam_amd.synth import synth_stereo_pair                  # noqa: E402

CFG = {"c2": (480, 752, 30, 20), "c3": (376, 1241, 25, 60), "c5": (720, 1280, 20, 20)}
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h, w, tile, th = CFG[name]
g = orb.ORBExtractor(h, w, 1.2, 8, 9, 14, 7, th, None, tile, tile, max_batch=n)
imgs = np.stack([synth_stereo_pair(1 + i // 2, h, w)[i & 1] for i in range(n)])
dev = torch.from_numpy(imgs).cuda()
L = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * 16)()
for rep in range(3):
    g.extract_batch_device_async(dev.data_ptr(), h * w, w, n, keep=dev)
    g.sync()
    L.jsorb_debug_detect_timing(out)
t = [int(x) for x in out]
names = ["waves", "prologue + staging", "barrier after staging", "phase 1 early rejects + appends", "phase 2 ring passes", "barrier after phase 2",
         "phase 3 NMS + arg-max", "barrier after phase 3", "phase 4 decode", "compact form: plane build (zero, barrier, scatter, barrier)"]
tot = sum(t[1:10])
print("k_detect %s, %d images: %d waves, %.0f clocks per wave" % (name, n, t[0], tot / max(t[0], 1)))
for k in (1, 2, 3, 4, 5, 9, 6, 7, 8):
    print("  %-36s %6.1f %%   %8.0f clk per wave" % (names[k], 100.0 * t[k] / tot, t[k] / max(t[0], 1)))
