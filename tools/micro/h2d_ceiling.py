#!/usr/bin/env python3
"""PCIe ceiling of the host-streamed regime: pinned host -> device copies of the size bench.py's host_streamed leg moves per step
(2 x 256 images of 752x480), as one copy, in lane-sized chunks on one stream, and in chunks alternating over two streams."""
import time, torch
N = 2 * 256 * 752 * 480
h = torch.empty(N, dtype=torch.uint8).pin_memory()
d = torch.empty(N, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(chunks, streams, reps=20):
    c = N // chunks
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(chunks):
            with torch.cuda.stream(streams[i % len(streams)]):
                d[i * c:(i + 1) * c].copy_(h[i * c:(i + 1) * c], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return reps * N / dt / 1e9
for chunks, streams, name in ((1, [s1], "one copy"), (8, [s1], "8 chunks, one stream"), (8, [s1, s2], "8 chunks, two streams"), (32, [s1], "32 chunks, one stream"), (32, [s1, s2], "32 chunks, two streams")):
    run(chunks, streams, 3)
    print("%-26s %.1f GB/s" % (name, run(chunks, streams)), flush=True)
