#!/usr/bin/env python3
"""bench.py - stereo pairs/s of the MI355X-native ORB front-end + stereo matcher (libjsorb).

A "step" = one pass of the hot path over one batch of synthetic stereo pairs already resident in HBM:
extract(left batch) + extract(right batch) + stereo match of every pair (7+5 kernel launches for the whole batch).
Workload at N=1: BASELINE.json configs[1] - EuRoC-shaped 752x480, 8 levels, scale 1.2, yaml-faithful tile 30
(cap 3466 keypoints/image), th_FAST 20, N in [9,14] - `--pairs` stereo pairs per GPU per step (weak scaling: every
rank processes its own pairs; the only collective is an RCCL all_gather of per-pair keypoint counts).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the stream it runs on) and
`cpu_baseline` (the oracle - the only CPU implementation of this algorithm that exists - timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (H, W, levels, tile, th_fast, fx, bf)   - SURVEY.md section 8(d)
    "c1": (240, 320, 3, 15, 20, 435.2, 47.906),
    "c2": (480, 752, 8, 30, 20, 435.2, 47.906),        # EuRoC.yaml:15,32,96-115
    "c3": (376, 1241, 8, 25, 60, 718.86, 386.14),      # KITTI
    "c5": (720, 1280, 8, 20, 20, 435.2, 47.906),       # KAIST-VIO shaped
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def algo_bytes_per_pair(ex):
    """ALGO_BYTES = 6*P + 24*T (P = sum of level pixels, T = sum of tiles), per stereo pair."""
    P = sum(h * w for h, w in ex.level_dims())
    return 6 * P + 24 * ex.T, P, ex.T


def usable_cores():
    """host cores this process may actually use: affinity mask, capped by the cgroup CPU quota (cpu.max) when there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, host_pairs, budget_s=12.0):
    """The oracle (kind 'port': the reference has no CPU path), OpenMP over independent pairs on all host
    cores, built -O3 -march=native on this box, same workload, bounded to ~budget_s of wall time."""
    from oracle import pyoracle as po
    H, W, L, tile, th, fx, bf = cfg
    try:
        po.build(native=True)
        native = True
    except Exception:
        native = False
    cores = usable_cores()
    lefts = np.stack([p[0] for p in host_pairs])
    rights = np.stack([p[1] for p in host_pairs])
    kw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
    n1, t1 = po.bench_pairs(lefts, rights, bf / fx, bf, 2.0, 1, native=native, **kw)
    n, t = po.bench_pairs(lefts, rights, bf / fx, bf, budget_s, cores, native=native, **kw)
    return {"value": round(n / t, 2), "unit": "stereo pairs/s", "cores": cores, "kind": "port",
            "single_thread_value": round(n1 / t1, 2),
            "sample": "%d pairs (cycling %d unique synthetic pairs of the same workload) in %.1f s on %d OpenMP threads; "
                      "oracle built -O3 -march=native=%s" % (n, len(host_pairs), t, cores, native)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs per GPU per step")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra steps with per-kernel hipEvent timing")
    ap.add_argument("--single-stream", action="store_true", help="all handles share one HIP stream (clean per-kernel times)")
    ap.add_argument("--groups", type=int, default=0, help="the step's pairs are split over this many independent left/right handle pairs, "
                    "one HIP stream per pair: kernels of different stages then overlap on the GPU (+6-8 %% over one pair of handles); "
                    "0 = up to 4, keeping at least ~20 EuRoC-sized images per launch")
    args = ap.parse_args()

    import torch
    from jetson_slam_amd import orb
    from jetson_slam_amd.synth import synth_stereo_pair

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    if os.environ.get("JSORB_BENCH_SINGLE_DEVICE"):       # test hook: every rank on cuda:0 (exercises the N > 1 code path on a 1-GPU box)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JSORB_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for the test hook above
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.config]
    H, W, L, tile, th, fx, bf = cfg
    P = args.pairs
    # synthetic EuRoC-shaped pairs; every rank gets its own seeds (independent pairs, no inter-GPU image traffic)
    n_unique = min(P, 16)
    host_pairs = [synth_stereo_pair(1 + rank * n_unique + i, H, W) for i in range(n_unique)]
    left_h = np.stack([host_pairs[i % n_unique][0] for i in range(P)])
    right_h = np.stack([host_pairs[i % n_unique][1] for i in range(P)])
    left_d = torch.from_numpy(left_h).to(dev)
    right_d = torch.from_numpy(right_h).to(dev)

    G = args.groups
    if G <= 0:      # auto: launches stay large enough to fill the GPU (at least ~7 Mpx, i.e. ~20 images of 752x480, per launch)
        G = max(1, min(4, int(P * H * W / 7.0e6)))
        while P % G:
            G -= 1
    if P % G:
        G = 1
    per = P // G                                  # pairs per handle pair and launch
    mk = lambda: orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, device_id=local_rank, max_batch=per)
    groups = [(mk(), mk()) for _ in range(G)]
    exl, exr = groups[0]
    handles = [h for pair in groups for h in pair]
    # The G handle pairs are independent of each other and run on G HIP streams (left and right of a pair share one): while one
    # pair is in its latency-bound stages (FAST ring test / NMS, descriptor gathers) another one is in a streaming stage.
    torch_stream_ptr = torch.cuda.current_stream(dev).cuda_stream
    shared_stream = None
    group_streams = []
    if args.single_stream:
        shared_stream = torch.cuda.Stream(dev)
        for h in handles:
            h.set_stream(shared_stream.cuda_stream)
    elif G > 1:
        for a, b in groups:
            st = torch.cuda.Stream(dev)
            group_streams.append(st)
            a.set_stream(st.cuda_stream)
            b.set_stream(st.cuda_stream)
    # payload of the collective, double-buffered: the gather kernels of step k+1 must not overwrite what the all_gather of step k reads
    counts_bufs = [torch.zeros(P * 3, dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [torch.zeros_like(counts_bufs[0]) for _ in range(world)] if world > 1 else None
    step_no = [0]
    mb = bf / fx

    def step():
        for gi, (a, b) in enumerate(groups):
            a.extract_batch_device_async(left_d[gi * per:].data_ptr(), H * W, W, per, keep=left_d)
            b.extract_batch_device_async(right_d[gi * per:].data_ptr(), H * W, W, per, keep=right_d)
        for a, b in groups:
            orb.stereo_match_batch_async(a, b, mb, bf)
        if world > 1:   # the one collective of the path: per-pair (N_left, N_right, N_matched), <1 KB per rank
            counts_d = counts_bufs[step_no[0] & 1]
            step_no[0] += 1
            for gi, (a, b) in enumerate(groups):
                orb.gather_counts_async(a, b, counts_d[gi * per * 3:].data_ptr())
                a.stream_wait_done(torch_stream_ptr)    # RCCL is issued from torch's stream: order it after the left streams
            dist.all_gather(gathered, counts_d)

    def fence():
        for h in handles:
            h.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    pairs_per_s = args.steps * P * world / dt

    # ---- parity spot check of the timed configuration (pair 0 of this rank) against the oracle ----
    parity = None
    roof = None
    cpu = None
    if rank == 0:
        from oracle import pyoracle as po
        kw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
        ol, orr = po.OracleExtractor(**kw), po.OracleExtractor(**kw)
        ol.extract(left_h[0])
        orr.extract(right_h[0])
        ou, od, _ = po.stereo_match(ol, orr, mb, bf)
        u, d, st = orb.stereo_result(exl, 0)
        parity = bool(np.array_equal(exl.keypoints(0), ol.keypoints()) and np.array_equal(exl.descriptors(0), ol.descriptors())
                      and np.array_equal(exr.keypoints(0), orr.keypoints()) and np.array_equal(exr.descriptors(0), orr.descriptors())
                      and np.array_equal(u.view(np.uint32), ou.view(np.uint32)) and np.array_equal(d.view(np.uint32), od.view(np.uint32)))

    # ---- per-kernel hipEvent timing pass (serialises launches, so it is separate from the timed region) ----
    # both handles on ONE stream here, so that a kernel's event-to-event time is its own duration and not the overlap with the
    # other handle's kernels (the timed region above overlaps left and right on two streams)
    if shared_stream is None:
        shared_stream = torch.cuda.Stream(dev)
        for h in handles:
            h.set_stream(shared_stream.cuda_stream)
    for e in handles:
        e.reset_kernel_timing()
        e.enable_kernel_timing(True)
    for _ in range(args.profile_steps):
        step()
    fence()
    kt = {}
    for e in handles:
        for k, (ms, n) in e.kernel_times().items():
            a = kt.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
        e.enable_kernel_timing(False)

    if rank == 0:
        ab, Ppx, T = algo_bytes_per_pair(exl)
        per_step_ms = {k: v[0] / max(1, args.profile_steps) for k, v in kt.items()}
        dom = max(per_step_ms, key=per_step_ms.get)
        avg_ms = kt[dom][0] / max(1, kt[dom][1])
        # one launch of an extract-side kernel covers `per` images = per/2 stereo pairs; a stereo-side launch covers `per` pairs
        units = per if dom in ("k_stereo", "k_median") else per / 2.0
        achieved = ab * units / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.config, {}).get(dom)
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_pair": ab, "pairs_per_launch": units,
                "pipeline_achieved": round(ab * pairs_per_s / world / 1e9, 1),
                "pipeline_frac": round(ab * pairs_per_s / world / 1e9 / HBM_PEAK_GBS, 4),
                "kernel_ms_per_step": {k: round(v, 4) for k, v in per_step_ms.items()}}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, host_pairs)
        n0 = exl.n_keypoints(0)
        out = {
            "metric": "stereo pairs/s (FAST+ORB extract L+R + stereo match)", "value": round(pairs_per_s, 1), "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (+f32 orientation/blur)",
            "data": "synthetic",
            "config": {"workload": "EuRoC-shaped %dx%d stereo, %d levels, scale 1.2, tile %d (cap %d kp/image), th_FAST %d, N[9,14]"
                                   % (W, H, L, tile, T, th) if args.config == "c2" else args.config,
                       "name": args.config, "pairs_per_gpu_per_step": P, "handle_pairs": G, "pairs_per_launch": per, "keypoints_image0": n0,
                       "inputs": "device-resident u8",
                       "parallelism": "independent pairs sharded over %d GPU(s); RCCL all_gather of counts only" % world},
            "parity_vs_oracle": parity, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
