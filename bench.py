#!/usr/bin/env python3
"""bench.py - stereo pairs/s of the MI355X-native ORB front-end + stereo matcher (libjsorb).

A "step" = one pass of the hot path over one batch of synthetic stereo pairs already resident in HBM:
extract(left batch) + extract(right batch) + stereo match of every pair, through ONE left/right handle pair (the library splits
a large batch over its internal lanes = HIP streams).
Workload at N=1: BASELINE.json configs[1] - EuRoC-shaped 752x480, 8 levels, scale 1.2, yaml-faithful tile 30
(cap 3466 keypoints/image), th_FAST 20, N in [9,14] - `--pairs` stereo pairs per GPU per step (weak scaling: every
rank processes its own pairs; the only collective is an RCCL all_gather of per-pair keypoint counts).
`--pairs-total X` switches to strong scaling (X pairs per step over all GPUs; X = 64 on c2 is BASELINE.json configs[3], "C4").

`python bench.py --gpus N` launches the N ranks itself (torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous) when it
is not already running under a launcher; under a launcher WORLD_SIZE must equal --gpus.

Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize and reduced with MAX over ranks;
blocks are repeated until >= --min-time seconds have been measured and the MEDIAN block is reported (`ms_per_step` = median block / K).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the stream it runs on), `cpu_baseline` (the
oracle - the only CPU implementation of this algorithm that exists - timed on the host cores) and, at N=1, the two other regimes of
the north star: `host_streamed` (pinned host memory -> hipMemcpyAsync -> kernels, PCIe-inclusive) and `frame_latency_us` (one
stereo pair through the reference-shaped synchronous C++ API).
"""
import argparse
import json
import os

# libjsorb's own default (jsorb_api.hip, jsorb_runtime_defaults), set here as well because torch initialises the HIP runtime before the
# library is loaded: 16 hardware queues, so that the library's lane / upload / main streams do not share a queue.  The
# value in effect is reported in the JSON line ("env").
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import socket
import subprocess
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
WORKLOAD_NAMES = {
    "c1": "C1 320x240 stereo, 3 levels, tile 15",
    "c2": "C2 EuRoC-shaped 752x480 stereo, 8 levels, scale 1.2, tile 30, th_FAST 20, N[9,14]",
    "c3": "C3 KITTI-shaped 1241x376 stereo, 8 levels, tile 25, th_FAST 60",
    "c5": "C5 KAIST-VIO-shaped 1280x720 stereo, 8 levels, tile 20",
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


def cpu_model():
    """model string of the host CPU (the CPU baseline states core count AND model)"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(cfg, lefts, rights, budget_s=12.0):
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
    kw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
    n1, t1 = po.bench_pairs(lefts, rights, bf / fx, bf, 2.0, 1, native=native, **kw)
    n, t = po.bench_pairs(lefts, rights, bf / fx, bf, budget_s, cores, native=native, **kw)
    return {"value": round(n / t, 2), "unit": "stereo pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "host_logical_cpus": os.cpu_count(), "single_thread_value": round(n1 / t1, 2),
            "sample": "%d pairs (cycling %d unique synthetic pairs of the same workload) in %.1f s on %d OpenMP threads; "
                      "oracle built -O3 -march=native=%s" % (n, len(lefts), t, cores, native)}


def multi_self_check(per_rank, single_device=False, gloo=False):
    """What the first real multi-GPU record has to explain by itself (round-5 review): every rank on its OWN device (device index and PCI address
    differ), every rank bound to its GPU's NUMA node, the one collective cheap (mean all_gather < 100 us over RCCL), and no straggler (a rank whose
    median block exceeds the fastest rank's by more than 5 % is named with what sets it apart).  Returns {"ok": bool, "checks": {...}, "diagnosis": str}.
    Pure function of the gathered per-rank records: tests/test_round6_host_logic.py feeds it hand-made ones."""
    n = len(per_rank)
    checks, notes = {}, []
    if single_device:
        checks["distinct_devices"] = checks["distinct_pci"] = "skipped (JSORB_BENCH_SINGLE_DEVICE test hook: every rank on cuda:0)"
    else:
        checks["distinct_devices"] = len({r["device_index"] for r in per_rank}) == n
        checks["distinct_pci"] = len({r["pci"] for r in per_rank}) == n
        if checks["distinct_devices"] is not True:
            notes.append("ranks share a device index: %s" % [r["device_index"] for r in per_rank])
        if checks["distinct_pci"] is not True:
            notes.append("ranks share a PCI address: %s" % [r["pci"] for r in per_rank])
    unbound = [i for i, r in enumerate(per_rank) if not r.get("numa_bound")]
    checks["numa_bound"] = not unbound
    if unbound:
        notes.append("rank(s) %s not bound to their GPU's NUMA node (no NUMA information, or JSORB_NO_PLACEMENT)" % unbound)
    ag = [r["all_gather_ms_mean"] for r in per_rank]
    if gloo:
        checks["all_gather_under_100us"] = "skipped (gloo test backend: host round trip)"
    else:
        checks["all_gather_under_100us"] = max(ag) < 0.1
        if max(ag) >= 0.1:
            notes.append("all_gather mean %.3f ms on rank %d (>= 0.1 ms: a 768-byte payload should cost link latency only - check xGMI topology / RCCL transport)" % (max(ag), ag.index(max(ag))))
    blocks = [r["median_block_ms"] for r in per_rank]
    fastest = min(blocks)
    slow = [i for i, b_ in enumerate(blocks) if b_ > 1.05 * fastest]
    checks["no_straggler_over_5pct"] = not slow
    for i in slow:
        r = per_rank[i]
        why = []
        if not r.get("numa_bound"):
            why.append("not NUMA-bound")
        if r["all_gather_ms_mean"] > 2 * min(ag) + 0.02:
            why.append("slow all_gather (%.3f ms)" % r["all_gather_ms_mean"])
        same_node = [j for j, q in enumerate(per_rank) if j != i and q.get("numa_node") == r.get("numa_node")]
        notes.append("rank %d (device %s, pci %s, numa %s) median block %.3f ms = +%.1f %% over the fastest rank%s%s" % (
            i, r["device_index"], r["pci"], r.get("numa_node"), blocks[i], 100.0 * (blocks[i] / fastest - 1.0),
            (": " + ", ".join(why)) if why else ": same binding and collective time as the others - look at the GPU itself (clocks, another tenant)",
            ("; shares NUMA node with rank(s) %s" % same_node) if same_node else ""))
    ok = all(v is True or isinstance(v, str) for v in checks.values())
    return {"ok": ok, "checks": checks, "diagnosis": "; ".join(notes) if notes else "all ranks within 5 % of the fastest, one device and one NUMA binding each, collective cheap"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run and become them."""
    import torch
    if not os.environ.get("JSORB_BENCH_SINGLE_DEVICE"):
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, n_dev))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs per GPU per step (weak scaling)")
    ap.add_argument("--pairs-total", type=int, default=0, help="strong scaling: this many pairs per step over ALL GPUs (64 on c2 = BASELINE C4)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--min-time", type=float, default=2.0, help="repeat the block of --steps steps until this many seconds are measured; the median block is reported")
    ap.add_argument("--tile", type=int, default=0, help="override the configuration's tile size (default: the yaml-faithful one)")
    ap.add_argument("--seed-base", type=int, default=1, help="seed of the first synthetic pair (parity soaks over other pairs: tools/micro/soak_bench.sh)")
    ap.add_argument("--unique", type=int, default=0, help="unique synthetic pairs per rank and input set (default: one per pair of the step, at most 128)")
    ap.add_argument("--input-sets", type=int, default=0, help="input sets rotated through the steps (default: 4 = 370 MB of level-0 data per GPU at the default c2 step - more than "
                    "the 256 MiB Infinity Cache, so that no step re-reads its images from the cache; 1 for the other configurations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-streamed / frame-latency / C4 side measurements")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra steps with per-kernel hipEvent timing")
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: one lane, all handles on ONE HIP stream (clean per-kernel durations under rocprofv3)")
    ap.add_argument("--dry-multi", action="store_true", help="N > 1 self-check only: a short run (no CPU baseline, no side measurements) whose line carries multi_gpu_diag.self_check - "
                    "distinct devices / PCI addresses, NUMA binding, all_gather < 100 us, no rank > 5 %% slower than the fastest; the ranks exit with 3 if a check fails (a launcher reports that as a failed worker)")
    ap.add_argument("--groups", type=int, default=1, help="split the step's pairs over this many independent left/right handle pairs on their own "
                    "HIP streams (round-1 schedule; the library now does the equivalent split internally, so the default is ONE handle pair)")
    args = ap.parse_args()
    if args.dry_multi:
        args.no_cpu_baseline = args.no_extras = True
        args.min_time = min(args.min_time, 0.5)
        args.profile_steps = 0

    if args.single_stream:
        os.environ["JSORB_MAX_LANES"] = "1"
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d - launch with --nproc-per-node equal to --gpus (or run plain "
                         "`python bench.py --gpus N`, which starts the ranks itself)" % (args.gpus, world))

    import torch
    from jetson_slam_amd import orb
    from jetson_slam_amd.synth import synth_stereo_pair

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    if os.environ.get("JSORB_BENCH_SINGLE_DEVICE"):       # test hook: every rank on cuda:0 (exercises the N > 1 code path on a 1-GPU box)
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # Rank placement (jetson_slam_amd/placement.py): the rank binds itself to the cores of its GPU's NUMA node BEFORE it allocates pinned host
    # memory or starts threads (first touch puts the pages there; 8 ranks x ~52 GB/s of pinned reads would otherwise cross the socket link for
    # half of the ranks).  At N > 1 right here; at N = 1 only around the host-streamed leg, so that the CPU baseline keeps every core it is granted.
    placement_info = [None]

    def apply_placement():
        if placement_info[0] is not None or os.environ.get("JSORB_NO_PLACEMENT"):
            return placement_info[0]
        from jetson_slam_amd import placement
        try:
            n_dev = torch.cuda.device_count()
            single = bool(os.environ.get("JSORB_BENCH_SINGLE_DEVICE"))
            addrs = []
            for r_ in range(world):
                pr = torch.cuda.get_device_properties(0 if single else min(r_, n_dev - 1))
                addrs.append(placement.pci_address(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
            n_share, k_share = placement.ranks_sharing_node(addrs)[rank % world]
            info = placement.bind(placement.plan(addrs[rank % world], sorted(os.sched_getaffinity(0)), k_share, n_share))
        except Exception as e:                         # never let placement break a run
            info = {"bound": False, "note": "placement failed: %s" % str(e)[:120]}
        placement_info[0] = info
        return info

    if world > 1:
        apply_placement()
    dist = None
    rccl_init_s = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JSORB_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for the test hook above
        t_init = time.perf_counter()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(8, dtype=torch.int32, device=dev)     # communicator set-up (rings over xGMI) happens at the first collective:
            dist.all_reduce(warm)                                    # done here, outside every timed region, and reported
            torch.cuda.synchronize(dev)
        else:
            dist.init_process_group(backend)
        rccl_init_s = time.perf_counter() - t_init
    gloo = world > 1 and os.environ.get("JSORB_BENCH_BACKEND", "nccl") != "nccl"

    cfg = CONFIGS[args.config]
    if args.tile > 0:           # the "nominal feature count" variants of SURVEY 8(d): c2 tile 58 (cap 999), c3 tile 46 (cap 2016), c5 tile 52 (cap 2920)
        cfg = (cfg[0], cfg[1], cfg[2], args.tile) + tuple(cfg[4:])
    H, W, L, tile, th, fx, bf = cfg
    strong = args.pairs_total > 0
    if strong:
        if args.pairs_total % world:
            raise SystemExit("--pairs-total %d is not divisible by %d GPUs" % (args.pairs_total, world))
        P = args.pairs_total // world
    else:
        P = args.pairs
    # synthetic EuRoC-shaped pairs; every rank gets its own seeds (independent pairs, no inter-GPU image traffic); every pair of a
    # step is a different image pair (up to 128 unique per rank)
    n_unique = args.unique if args.unique > 0 else min(P, 128)
    n_unique = max(1, min(n_unique, P))
    # ONE global list of P * world pairs per step, cut into contiguous blocks (jetson_slam_amd.batch.shard_range): pair p of the step is the
    # synthetic pair with seed seed_base + p and belongs to the rank whose block holds p
    from jetson_slam_amd.batch import shard_range
    p_first, p_end = shard_range(P * world, rank, world)
    assert p_end - p_first == P
    # input sets: step k works on set k mod n_sets (different synthetic pairs in every set), so that consecutive steps do not find their
    # images in the 256 MiB Infinity Cache
    n_sets = args.input_sets if args.input_sets > 0 else (4 if (args.config == "c2" and args.tile <= 0 and not args.single_stream and not strong and P >= 64) else 1)
    sets_u, sets_d = [], []
    for si in range(n_sets):
        host_pairs = [synth_stereo_pair(args.seed_base + si * P * world + p_first + i, H, W) for i in range(n_unique)]
        lu, ru = np.stack([p[0] for p in host_pairs]), np.stack([p[1] for p in host_pairs])
        idx = np.arange(P) % n_unique
        sets_u.append((lu, ru))
        sets_d.append((torch.from_numpy(lu[idx]).to(dev), torch.from_numpy(ru[idx]).to(dev)))
    left_u, right_u = sets_u[0]
    left_d, right_d = sets_d[0]
    cur_set = [0]

    G = max(1, args.groups)
    while P % G:
        G -= 1
    per = P // G                                  # pairs per handle pair
    mk = lambda b=per: orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, device_id=local_rank, max_batch=b)
    groups = [(mk(), mk()) for _ in range(G)]
    exl, exr = groups[0]
    handles = [h for pair in groups for h in pair]
    torch_stream = torch.cuda.current_stream(dev)
    group_streams = []
    if args.single_stream:
        one = torch.cuda.Stream(dev)
        group_streams.append(one)
        for h in handles:
            h.set_stream(one.cuda_stream)
    elif G > 1:                                   # round-1 schedule: one HIP stream per handle pair
        for a, b in groups:
            st = torch.cuda.Stream(dev)
            group_streams.append(st)
            a.set_stream(st.cuda_stream)
            b.set_stream(st.cuda_stream)
    main_streams = [torch.cuda.ExternalStream(a._lib.jsorb_get_stream(a.handle), device=dev) for a, _ in groups]
    # payload of the collective, double-buffered: the gather kernels of step k+2 must not overwrite what the all_gather of step k reads
    counts_bufs = [torch.zeros(P * 3, dtype=torch.int32, device=dev) for _ in range(2)]
    counts_free = [None, None]                    # event recorded on torch's stream after the all_gather that read the buffer
    gathered = [torch.zeros_like(counts_bufs[0]) for _ in range(world)] if world > 1 else None
    step_no = [0]
    mb = bf / fx
    ag_probe = [None]                             # diagnostic steps: a list that collects (start, end) of every all_gather (events on torch's stream; wall clock under gloo)

    def step(which=None):
        si = cur_set[0] if which is None else which
        cur_set[0] = (si + 1) % n_sets
        ld, rd = sets_d[si]
        for gi, (a, b) in enumerate(groups):
            a.extract_batch_device_async(ld[gi * per:].data_ptr(), H * W, W, per, keep=ld)
            b.extract_batch_device_async(rd[gi * per:].data_ptr(), H * W, W, per, keep=rd)
        for a, b in groups:
            orb.stereo_match_batch_async(a, b, mb, bf)
        if world > 1:   # the one collective of the path: per-pair (N_left, N_right, N_matched), <2 KB per rank
            k = step_no[0] & 1
            step_no[0] += 1
            counts_d = counts_bufs[k]
            for gi, (a, b) in enumerate(groups):
                if counts_free[k] is not None:
                    main_streams[gi].wait_event(counts_free[k])          # the all_gather that last read this buffer has finished
                orb.gather_counts_async(a, b, counts_d[gi * per * 3:].data_ptr())
                a.stream_wait_done(torch_stream.cuda_stream)            # the collective is issued from torch's stream: order it after the handles
            if gloo:
                host = counts_d.cpu()
                parts = [torch.zeros_like(host) for _ in range(world)]
                t_ag = time.perf_counter()
                dist.all_gather(parts, host)
                if ag_probe[0] is not None:
                    ag_probe[0].append((t_ag, time.perf_counter()))
                for g_, p_ in zip(gathered, parts):
                    g_.copy_(p_)
            else:
                if ag_probe[0] is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(torch_stream)
                dist.all_gather(gathered, counts_d)
                if ag_probe[0] is not None:
                    e1.record(torch_stream)
                    ag_probe[0].append((e0, e1))
            ev = torch.cuda.Event()
            ev.record(torch_stream)
            counts_free[k] = ev

    def fence():
        for h in handles:
            h.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed_block(n_steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        blocks_local.append(dt)                   # this rank's own clock (the line reports every rank's median next to the max the value is made of)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    blocks_local = []

    for _ in range(args.warmup):
        step()
    blocks = [timed_block(args.steps)]
    # every rank must run the same number of blocks: rank 0 decides from its first block
    n_blocks = int(min(200, max(3 if args.min_time > 0 else 1, np.ceil(args.min_time / max(blocks[0], 1e-6)))))
    if world > 1:
        nb = torch.tensor([n_blocks], dtype=torch.int64, device="cpu" if gloo else dev)
        dist.broadcast(nb, 0)
        n_blocks = int(nb.item())
    for _ in range(n_blocks - 1):
        blocks.append(timed_block(args.steps))
    dt = median(blocks)
    pairs_per_s = args.steps * P * world / dt

    # ---- parity of EVERY unique pair of this rank against the oracle: counts + a checksum of kp | desc | kp | desc | uRight | depth ----
    from oracle import pyoracle as po
    okw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
    cores = usable_cores()
    bound = bool(placement_info[0] and placement_info[0].get("bound"))      # a bound rank's affinity mask already is its share of the host
    n_bad = 0
    for si in range(n_sets):                 # every input set once more (untimed), every unique pair of it against the oracle; the last set stays in the handles
        step(si)
        fence()
        o_digest, o_counts = po.pairs_digest(sets_u[si][0], sets_u[si][1], mb, bf, max(1, cores if bound else cores // world), **okw)
        for i in range(n_unique):
            a, b = groups[i // per]
            j = i % per
            u, d, st = orb.stereo_result(a, j)
            got = po.digest_arrays([a.keypoints(j), a.descriptors(j), b.keypoints(j), b.descriptors(j), u, d])
            if got != int(o_digest[i]) or [a.n_keypoints(j), b.n_keypoints(j), st["n_final"]] != o_counts[i].tolist():
                n_bad += 1
    parity_local = n_bad == 0
    counts_ok = True
    if world > 1:
        # EVERY rank checks EVERY row of the table the last step's all_gather delivered to it: the expected table is the oracle's counts of
        # all P * world pairs in global pair order (each rank's oracle computes its own block from the seeds; the blocks are exchanged with a
        # second, untimed all_gather)
        torch.cuda.synchronize(dev)
        exp_local = torch.from_numpy(np.ascontiguousarray(o_counts[np.arange(P) % n_unique], dtype=np.int32)).reshape(-1)
        if gloo:
            exp_parts = [torch.zeros_like(exp_local) for _ in range(world)]
            dist.all_gather(exp_parts, exp_local)
        else:
            exp_dev = exp_local.to(dev)
            exp_parts = [torch.zeros_like(exp_dev) for _ in range(world)]
            dist.all_gather(exp_parts, exp_dev)
        counts_ok = all(bool(torch.equal(gathered[r_].cpu(), exp_parts[r_].cpu())) for r_ in range(world))
        flag = torch.tensor([1 if (parity_local and counts_ok) else 0], dtype=torch.int32, device="cpu" if gloo else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity = bool(flag.item())
    else:
        parity = parity_local

    # ---- what makes the first real multi-GPU run diagnosable: every rank's own median block time, and the time of the all_gather itself ----
    multi_diag = None
    if world > 1:
        ag_probe[0] = []
        for _ in range(12):
            step()
        fence()
        if gloo:
            ag_ms = [(b - a) * 1e3 for a, b in ag_probe[0]]
        else:
            ag_ms = [a.elapsed_time(b) for a, b in ag_probe[0]]
        ag_probe[0] = None
        props_ = torch.cuda.get_device_properties(dev)
        mine = {"median_block_ms": round(median(blocks_local) * 1e3, 3), "ms_per_step": round(median(blocks_local) / args.steps * 1e3, 4),
                "all_gather_ms_mean": round(float(np.mean(ag_ms[2:])), 4), "all_gather_ms_max": round(float(np.max(ag_ms[2:])), 4),
                "device_index": int(dev.index or 0), "pci": "%04x:%02x:%02x.0" % (props_.pci_domain_id, props_.pci_bus_id, props_.pci_device_id),
                "numa_bound": bool(placement_info[0] and placement_info[0].get("bound")), "numa_node": (placement_info[0] or {}).get("numa_node"),
                "pid": os.getpid()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        multi_diag = {"per_rank": per_rank, "all_gather_clock": "wall clock around dist.all_gather (gloo)" if gloo else "events on torch's stream around dist.all_gather (RCCL)",
                      "note": "value is made of the MAX over ranks of each block; per_rank holds every rank's own median",
                      "self_check": multi_self_check(per_rank, single_device=bool(os.environ.get("JSORB_BENCH_SINGLE_DEVICE")), gloo=gloo)}

    # ---- BASELINE C4 shape as a side measurement: 64 pairs per iteration over all GPUs, one all_gather per iteration ----
    c4 = None
    if args.config == "c2" and not strong and not args.no_extras and 64 % world == 0 and 64 // world <= per and G == 1:
        p4 = 64 // world
        saved = (left_d, right_d)

        def step_c4():
            exl.extract_batch_device_async(left_d.data_ptr(), H * W, W, p4, keep=left_d)
            exr.extract_batch_device_async(right_d.data_ptr(), H * W, W, p4, keep=right_d)
            orb.stereo_match_batch_async(exl, exr, mb, bf)
            if world > 1:
                orb.gather_counts_async(exl, exr, counts_bufs[0].data_ptr())
                exl.stream_wait_done(torch_stream.cuda_stream)
                if gloo:
                    host = counts_bufs[0][:p4 * 3].cpu()
                    dist.all_gather([torch.zeros_like(host) for _ in range(world)], host)
                else:
                    dist.all_gather([g_[:p4 * 3] for g_ in gathered], counts_bufs[0][:p4 * 3])
                main_streams[0].wait_stream(torch_stream)
        for _ in range(5):
            step_c4()
        n4 = 60
        fence()
        t0 = time.perf_counter()
        for _ in range(n4):
            step_c4()
        fence()
        d4 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d4], dtype=torch.float64, device="cpu" if gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d4 = float(t.item())
        c4 = {"workload": "BASELINE C4: 64 EuRoC-shaped pairs per iteration sharded over %d GPU(s) (%d per GPU), all_gather of counts per iteration" % (world, p4),
              "value": round(64 * n4 / d4, 1), "unit": "stereo pairs/s", "ms_per_iter": round(d4 / n4 * 1e3, 4), "iters": n4, "scaling": "strong"}
        del saved

    # ---- per-kernel hipEvent timing pass (serialises launches and forces one lane, so it is separate from the timed region) ----
    # all handles on ONE stream here, so that a kernel's event-to-event time is its own duration and not the overlap with the other
    # handle's kernels (the timed region above overlaps left and right on different streams)
    shared_stream = torch.cuda.Stream(dev)
    for e in handles:
        e.set_stream(shared_stream.cuda_stream)
        e.enable_kernel_timing(True)
    kt = timed_steps(handles, step, fence, args.profile_steps)
    for e in handles:
        e.enable_kernel_timing(False)

    # ---- north-star regime (i) at N > 1: every rank streams its own pairs from its own pinned host buffers at the same time (per-GPU PCIe
    # links; no inter-GPU traffic); rank 0 reports the sum and the per-GPU rates ----
    hs_multi = None
    if world > 1 and not args.no_extras and args.config == "c2" and not strong and args.tile <= 0:
        dist.barrier()
        hs = measure_host_streamed(orb, torch, cfg, left_u, right_u, dev)
        mine = torch.tensor([hs["value"], hs["pcie_gb_per_s"]], dtype=torch.float64, device="cpu" if gloo else dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        vals = [[float(x) for x in q.cpu()] for q in parts]
        hs_multi = {"value": round(sum(v[0] for v in vals), 1), "unit": "stereo pairs/s", "per_gpu": [round(v[0], 1) for v in vals],
                    "pcie_gb_per_s_per_gpu": [round(v[1], 2) for v in vals], "pcie_gb_per_s": round(sum(v[1] for v in vals), 2),
                    "sample": "all %d ranks at the same time, each: %s" % (world, hs["sample"])}
    placements = None
    if world > 1:                                       # what every rank bound itself to (rank order)
        placements = [None] * world
        dist.all_gather_object(placements, placement_info[0])
    if rank == 0:
        ab, Ppx, T = algo_bytes_per_pair(exl)
        per_step_ms = {k: v[2] for k, v in kt.items()}
        if args.profile_steps <= 0 or not any(v[1] for v in kt.values()):
            kt = {"k_detect": [1.0, 1]}                 # no profiling pass requested: the roofline block is a placeholder
            per_step_ms = {"k_detect": 0.0}
        dom = max(per_step_ms, key=per_step_ms.get)
        avg_ms = kt[dom][0] / max(1, kt[dom][1])
        # one launch of an extract-side kernel covers `per` images = per/2 stereo pairs; a stereo-side launch covers `per` pairs
        units = per if dom in ("k_stereo", "k_median") else per / 2.0
        achieved = ab * units / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source, step_traffic = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                per_kernel_traffic = tj.get(args.config if args.tile <= 0 else "", {})
                traffic = per_kernel_traffic.get(dom)
                traffic_source = tj.get("_source", "profiles/hbm_traffic.json") + " (builder-measured rocprofv3 PMC pass, NOT measured in this run)"
                # HBM bytes of one step: every extract-side kernel runs twice (left and right images), the stereo-side ones once
                if per_kernel_traffic and per == {"c2": 128}.get(args.config, 64):      # (the profile passes ran 128 images per launch at c2, 64 at c3 / c5: tools/profile_round.sh)
                    step_traffic = sum((1 if k in ("k_stereo", "k_median") else 2) * v for k, v in per_kernel_traffic.items() if k.startswith("k_") and isinstance(v, (int, float)))
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "bound_note": "the contract's roofline is HBM bandwidth; what binds this integer / bitwise path is vector-instruction issue (bound_actual, valu_issue_frac).  "
                                               "frac is ALGORITHMIC bytes of the whole pipeline over the dominant kernel's launch time (the contract's convention): it moves with k_detect's duration only - "
                                               "0.158 / 0.192 / 0.223 / 0.202 in rounds 1-4 (round 4 fell because k_detect's LDS request was raised on purpose, to leave room for a k_describe "
                                               "workgroup per CU), ~0.26 with the compact k_detect of round 5.  kernel_own_hbm_frac is what that kernel itself moves over HBM", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_pair": ab, "pairs_per_launch": units,
                "pipeline_achieved": round(ab * pairs_per_s / world / 1e9, 1),
                "pipeline_frac": round(ab * pairs_per_s / world / 1e9 / HBM_PEAK_GBS, 4),
                "kernel_ms_per_step": {k: round(v, 4) for k, v in per_step_ms.items()}}
        # SURVEY 8(d) left the data-dependent bytes out of ALGO_BYTES as "< 6 %": outputs 56 N per image and, for the matcher, 64 N + 12 C + 462 M (N left
        # keypoints, C candidate pairs, M window searches).  On the benchmark images they are a THIRD of the figure (round-5 review) - reported here as a
        # second figure from the counts of the step's first pair; `frac` keeps the input-independent convention.
        try:
            _u0, _d0, st0 = orb.stereo_result(exl, 0)
            nl0, nr0 = exl.n_keypoints(0), exr.n_keypoints(0)
            extra = 56 * (nl0 + nr0) + 64 * nl0 + 12 * int(st0["n_candidate_pairs"]) + 462 * int(st0["n_corr_match"])
            roof["algo_bytes_per_pair_incl_outputs_and_stereo"] = int(ab + extra)
            roof["data_dependent_share"] = round(extra / float(ab + extra), 3)
        except Exception:
            pass
        if traffic:
            roof["kernel_own_hbm_frac"] = round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)      # the dominant kernel's own PMC bytes / its duration / peak
        if step_traffic:
            roof["step_traffic_bytes"] = int(step_traffic)
            roof["step_traffic_over_algorithmic"] = round(step_traffic / (ab * P), 3)
            roof["step_traffic_frac"] = round(step_traffic / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)      # all HBM bytes of a step / step time / peak
        props = torch.cuda.get_device_properties(dev)
        vv = valu_view(args.config if args.tile <= 0 else "", pairs_per_s / world, props.multi_processor_count, props.clock_rate * 1e3 if getattr(props, "clock_rate", 0) else 2.4e9)
        if vv:
            roof.update(vv)
            # the two clocks side by side (round-5 review): `avg_launch_ms` above is this run's hipEvent time of the dominant kernel on its own stream (the
            # --profile-steps pass), this one the builder's rocprofv3 --kernel-trace average of the same launch shape (profiles/valu_counters.json, digest-checked)
            try:
                tj_ = json.load(open(os.path.join(ROOT, "profiles", "valu_counters.json")))
                us_ = tj_[args.config]["avg_us"].get("k_compact_flat" if dom == "k_compact" else dom) if args.tile <= 0 and per == int(tj_[args.config]["_pairs_per_launch"]) else None
                if us_:
                    roof["avg_launch_ms_rocprofv3"] = round(us_ / 1e3, 4)
                    roof["frac_with_rocprofv3_duration"] = round(ab * units / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            except Exception:
                pass
        if not args.no_extras:
            pm = measure_copy_peak(torch, dev)
            roof["peak_measured"] = pm
            roof["peak_measured_what"] = "1 GiB device-to-device copy, read + write bytes, best of 6 (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)"
            roof["frac_of_measured_peak"] = round(achieved / pm, 4)
            roof["pipeline_frac_of_measured_peak"] = round(ab * pairs_per_s / world / 1e9 / pm, 4)
            if step_traffic:
                roof["step_traffic_frac_of_measured_peak"] = round(step_traffic / (dt / args.steps) / 1e9 / pm, 4)
        cpu = None
        host_streamed = hs_multi
        frame_latency = None
        other = None
        other_inputs = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(cfg, left_u[:16], right_u[:16])
        if world == 1 and not args.no_extras:
            # the device-resident handles stay alive: handles created right after others of the process were destroyed measured 17 %
            # less in this regime (57.9 k against 70 k pairs/s, tools/micro/hs_mimic.py dev_closed / dev_closelate - device memory handed
            # back to the runtime and allocated again), which is a property of the allocation history, not of the regime
            saved_affinity = os.sched_getaffinity(0)    # N = 1: the binding holds around the host-streamed leg ONLY (round-4 review: it used to stay
            apply_placement()                           # for every later leg); pinned buffers of the leg are allocated by a thread on the GPU's NUMA node
            try:
                host_streamed = measure_host_streamed(orb, torch, cfg, left_u, right_u, dev)
            finally:
                try:
                    os.sched_setaffinity(0, saved_affinity)
                except OSError:
                    pass
            for h in handles:
                h.close()
            try:                                        # the C++ driver's threads inherit the binding: same NUMA node as the GPU (round-5 review)
                if placement_info[0] and placement_info[0].get("bound") and placement_info[0].get("cpus"):
                    from jetson_slam_amd import placement
                    os.sched_setaffinity(0, set(placement.parse_cpulist(placement_info[0]["cpus"])))
                frame_latency = measure_frame_latency(cfg, left_u[:4], right_u[:4])
            finally:
                try:
                    os.sched_setaffinity(0, saved_affinity)
                except OSError:
                    pass
            if args.config == "c2" and args.tile <= 0 and not strong:
                # BASELINE C3 / C5 and the "nominal feature count" tiles of SURVEY 8(d), ~1 s each, every unique pair checked
                other = {}
                for key, (nm, tl, pp, nu) in {"c3": ("c3", 0, 64, 16), "c5": ("c5", 0, 64, 8), "c2_tile58": ("c2", 58, 128, 16),
                                              "c3_tile46": ("c3", 46, 64, 16), "c5_tile52": ("c5", 52, 64, 8)}.items():
                    try:
                        other[key] = measure_other_config(orb, torch, dev, nm, tl, pp, nu)
                    except Exception as e:      # never let a side measurement break the contract line
                        other[key] = {"error": str(e)[:200]}
                # other image statistics at the headline geometry (round-5 review: every list / pool size was tuned on ONE generator): C2, batch 128,
                # 16 unique pairs of each family checked against the oracle
                from jetson_slam_amd.synth import INPUT_FAMILIES
                other_inputs = {}
                for fam in INPUT_FAMILIES:
                    try:
                        other_inputs[fam] = measure_other_config(orb, torch, dev, "c2", 0, 128, 16, family=fam)
                        other_inputs[fam]["frac_of_headline"] = round(other_inputs[fam]["value"] / pairs_per_s, 3)
                    except Exception as e:
                        other_inputs[fam] = {"error": str(e)[:200]}
        n0 = int(o_counts[0][0])
        out = {
            "metric": "stereo pairs/s (FAST+ORB extract L+R + stereo match)", "value": round(pairs_per_s, 1), "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8/int32 (+f32 orientation/blur)",
            "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.config] + (" [tile overridden: %d]" % args.tile if args.tile > 0 else "") + (" (cap %d kp/image)" % T) +
                                   ("; BASELINE C4: %d pairs per step sharded over %d GPU(s)" % (args.pairs_total, world) if strong else ""),
                       "name": args.config, "pairs_per_gpu_per_step": P, "pairs_per_step_total": P * world, "unique_pairs_per_gpu": n_unique * n_sets, "input_sets": n_sets,
                       "handle_pairs": G, "library_lanes": "up to 4 HIP streams per handle (>= ~7 Mpx per lane)", "keypoints_image0": n0,
                       "inputs": "device-resident u8; %d input set(s) of %d pairs rotated step by step = %.0f MB of level-0 images per GPU (Infinity Cache: 256 MiB)"
                                 % (n_sets, P, n_sets * P * 2 * H * W / 1e6),
                       "parallelism": "independent pairs sharded over %d GPU(s); RCCL all_gather of counts only" % world},
            "timing": {"blocks": len(blocks), "block_steps": args.steps, "measured_s": round(sum(blocks), 3), "statistic": "median block, max over ranks",
                       "block_ms_min": round(min(blocks) * 1e3, 3), "block_ms_max": round(max(blocks) * 1e3, 3)},
            "parity_vs_oracle": parity, "parity_pairs_checked": n_unique * n_sets * world, "gathered_counts_ok": counts_ok if world > 1 else None,
            "roofline": roof, "cpu_baseline": cpu, "host_streamed": host_streamed, "frame_latency_us": frame_latency, "c4_batch64": c4, "multi_gpu_diag": multi_diag,
            "other_configs": other, "other_inputs": other_inputs, "rccl_init_s": None if rccl_init_s is None else round(rccl_init_s, 3),
            "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "placement": placements if world > 1 else [placement_info[0]],
        }
        # the numbers a reader looks for first, once more at the END of the line (a log that keeps only the tail of the line still shows them)
        def _v(d, *ks):
            for k_ in ks:
                d = d.get(k_) if isinstance(d, dict) else None
            return d
        out["summary"] = {"pairs_per_s": out["value"], "ms_per_step": out["ms_per_step"], "parity_vs_oracle": parity, "roofline_frac": roof.get("frac"),
                          "dominant_kernel": dom, "dominant_kernel_ms": roof.get("avg_launch_ms"), "kernel_own_hbm_frac": roof.get("kernel_own_hbm_frac"),
                          "step_traffic_frac_of_measured_peak": roof.get("step_traffic_frac_of_measured_peak"), "valu_issue_frac": roof.get("valu_issue_frac"),
                          "cpu_baseline_pairs_per_s": _v(cpu, "value"), "host_streamed_pairs_per_s": _v(host_streamed, "value"),
                          "frame_latency_us_median": _v(frame_latency, "total_us_median"), "frame_latency_us_p90": _v(frame_latency, "total_us_p90"),
                          "frame_latency_us_median_persistent_threads": _v(frame_latency, "total_us_median_persistent_threads"),
                          "c4_batch64_pairs_per_s": _v(c4, "value"),
                          "other_configs_pairs_per_s": None if not other else {k_: _v(v_, "value") for k_, v_ in other.items()},
                          "other_inputs_pairs_per_s": None if not other_inputs else {k_: _v(v_, "value") for k_, v_ in other_inputs.items()}}
        print(json.dumps(out), flush=True)
        if world > 1 and multi_diag and not multi_diag["self_check"]["ok"]:
            print("[bench.py multi-GPU self-check] " + multi_diag["self_check"]["diagnosis"], file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        if args.dry_multi and multi_diag and not multi_diag["self_check"]["ok"]:
            sys.exit(3)


def measure_copy_peak(torch, dev, gib=1.0, reps=6):
    """"confirm on the box with a copy kernel and state the measured peak" - a 1 GiB device-to-device copy (16 B per lane on
    both sides), read + write bytes over the best of `reps` event-timed runs."""
    n = int(gib * (1 << 30)) // 16
    src = torch.empty((n, 4), dtype=torch.int32, device=dev).fill_(1)
    dst = torch.empty_like(src)
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src)
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    del src, dst
    torch.cuda.empty_cache()
    return round(2.0 * n * 16 / (best * 1e-3) / 1e9, 1)


def valu_view(config, pairs_per_s, n_cus, clock_hz):
    """What actually bounds the path: vector-ALU issue (roofline.bound_actual).  Per kernel launch, SQ_INSTS_VALU and the stand-alone launch duration come
    from the builder's rocprofv3 passes (profiles/valu_counters.json), the mean issue cost of a kernel's vector instructions from its assembly
    (profiles/valu_mix.json: plain 32-bit arithmetic issues every 2 clocks per SIMD, packed / permute / convert / compare instructions every 4,
    profiles/r04_valu_rate.txt).  Both files carry the digest of the kernel sources they were made from: if it is not the digest of THIS build, nothing
    is reported.  NOT measured in this run - the run contributes pairs/s only."""
    try:
        from jetson_slam_amd import build as jb
        tj = json.load(open(os.path.join(ROOT, "profiles", "valu_counters.json")))
        mj = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
        sha = jb.csrc_sha256()
        if tj.get("_csrc_sha256") != sha or mj.get("_csrc_sha256") != sha:
            return {"valu_note": "profiles/valu_counters.json / valu_mix.json were made from other kernel sources than this build: not reported"}
        k = tj[config]
        ppl = float(k["_pairs_per_launch"])
        simd_hz = n_cus * 4 * clock_hz                        # SIMD-clocks per second of the chip
        per_pair_instr, per_pair_clk4, per_pair_clkmix, per_kernel = 0.0, 0.0, 0.0, {}
        for side, mult in (("extract_side", 2.0), ("stereo_side", 1.0)):
            for name, instr in k[side].items():
                mix = mj["kernels"].get(name, {}).get("clk_per_valu_instr", 4.0)
                per_pair_instr += mult * instr / ppl
                per_pair_clk4 += mult * instr * 4.0 / ppl
                per_pair_clkmix += mult * instr * mix / ppl
                us = k.get("avg_us", {}).get(name)
                per_kernel[name] = {"valu_instr_per_launch": instr, "clk_per_valu_instr": mix,
                                    "own_valu_bound_frac": None if not us else round(instr * mix / (simd_hz * us * 1e-6), 3)}
        return {"bound_actual": "valu_issue",
                "valu_wave_instr_per_pair": round(per_pair_instr), "valu_wave_instr_per_128_pairs": round(per_pair_instr * 128),
                "valu_issue_frac": round(per_pair_clkmix * pairs_per_s / simd_hz, 4),
                "valu_issue_frac_all_4clk": round(per_pair_clk4 * pairs_per_s / simd_hz, 4),
                "valu_issue_peak": "%d CUs x 4 SIMDs x %.2f GHz; issue cost per instruction class from profiles/r04_valu_rate.txt (2 / 4 / 8 clk), per-kernel mix from "
                                   "profiles/valu_mix.json (static, loop-depth weighted); _all_4clk prices every instruction at 4 clk" % (n_cus, clock_hz / 1e9),
                "per_kernel": per_kernel,
                "valu_source": tj.get("_source", "profiles/valu_counters.json") + " (builder-measured rocprofv3 passes, NOT measured in this run; digest of the kernel sources checked)"}
    except Exception:
        return None


def timed_steps(handles, step, fence, n_steps):
    """Per-kernel hipEvent times of n_steps steps, ONE step at a time: {kernel: [median step's ms, its launches, that same ms]} - the median over
    the steps, not their mean: with event pairs around every launch the host is the slower side, the GPU runs dry between steps and the
    first kernels after an idle gap are sometimes timed at two or three times their duration (seen as k_pyramid or k_median 'dominating')."""
    per = {}
    for _ in range(max(0, n_steps)):
        for e in handles:
            e.reset_kernel_timing()
        step()
        fence()
        acc = {}
        for e in handles:
            for k, (ms, n) in e.kernel_times().items():
                x = acc.setdefault(k, [0.0, 0])
                x[0] += ms
                x[1] += n
        for k, (ms, n) in acc.items():
            per.setdefault(k, []).append((ms, n))
    out = {}
    for k, v in per.items():
        v.sort()
        ms, n = v[len(v) // 2]
        out[k] = [ms, n, ms]
    return out


def measure_other_config(orb, torch, dev, name, tile_override, P, n_unique, seconds=0.8, cache={}, family=None):
    """The other BASELINE configurations in the driver-run line: device-resident batches of P pairs through ONE handle pair, every unique pair
    checked against the oracle, per-kernel hipEvent pass for the dominant kernel's roofline fraction.
    family: another image statistic than synth_stereo_pair (jetson_slam_amd.synth.INPUT_FAMILIES) - the line's `other_inputs`."""
    from jetson_slam_amd.synth import synth_stereo_pair, synth_family_pair
    from oracle import pyoracle as po
    H, W, L, tile, th, fx, bf = CONFIGS[name]
    if tile_override:
        tile = tile_override
    if (name, family) not in cache:
        prs = [synth_family_pair(family, 1 + i, H, W) if family else synth_stereo_pair(1 + i, H, W) for i in range(n_unique)]
        cache[(name, family)] = (np.stack([q[0] for q in prs]), np.stack([q[1] for q in prs]))
    left_u, right_u = cache[(name, family)]
    idx = np.arange(P) % left_u.shape[0]
    left_d, right_d = torch.from_numpy(left_u[idx]).to(dev), torch.from_numpy(right_u[idx]).to(dev)
    mk = lambda: orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, device_id=dev.index or 0, max_batch=P)
    a, b = mk(), mk()
    mb = bf / fx

    def step():
        a.extract_batch_device_async(left_d.data_ptr(), H * W, W, P, keep=left_d)
        b.extract_batch_device_async(right_d.data_ptr(), H * W, W, P, keep=right_d)
        orb.stereo_match_batch_async(a, b, mb, bf)

    def fence():
        a.sync(); b.sync()
    for _ in range(3):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(4):
        step()
    fence()
    n = max(4, int(4 * seconds / max(time.perf_counter() - t0, 1e-6)))
    blocks = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        fence()
        blocks.append((time.perf_counter() - t0) / n)
    dt = median(blocks)
    okw = dict(height=H, width=W, n_levels=L, tile_h=tile, tile_w=tile, th_fast_max=th)
    o_digest, o_counts = po.pairs_digest(left_u, right_u, mb, bf, usable_cores(), **okw)
    bad = 0
    for i in range(left_u.shape[0]):
        u, d, st = orb.stereo_result(a, i)
        got = po.digest_arrays([a.keypoints(i), a.descriptors(i), b.keypoints(i), b.descriptors(i), u, d])
        if got != int(o_digest[i]) or [a.n_keypoints(i), b.n_keypoints(i), st["n_final"]] != o_counts[i].tolist():
            bad += 1
    one = torch.cuda.Stream(dev)
    for e in (a, b):
        e.set_stream(one.cuda_stream)
        e.enable_kernel_timing(True)
    kt = timed_steps((a, b), step, fence, 5)
    ab, Ppx, T = algo_bytes_per_pair(a)
    per_step = {k: v[2] for k, v in kt.items() if v[1]}
    dom = max(per_step, key=per_step.get)
    avg_ms = kt[dom][0] / kt[dom][1]
    units = P if dom in ("k_stereo", "k_median") else P / 2.0
    pps = P / dt
    out = {"workload": WORKLOAD_NAMES[name] + (" [tile overridden: %d]" % tile_override if tile_override else "") + " (cap %d kp/image)" % T,
           "value": round(pps, 1), "unit": "stereo pairs/s", "ms_per_step": round(dt * 1e3, 4), "pairs_per_step": P,
           "keypoints_image0": int(o_counts[0][0]), "parity_vs_oracle": bad == 0, "parity_pairs_checked": int(left_u.shape[0]),
           "algo_bytes_per_pair": ab, "pipeline_frac": round(ab * pps / 1e9 / HBM_PEAK_GBS, 4),
           "kernel": dom, "frac": round(ab * units / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4),
           "kernel_ms_per_step": {k: round(v, 4) for k, v in per_step.items()}}
    if family:
        out["workload"] += "; input family '%s' (jetson_slam_amd.synth.synth_family_pair)" % family
        out["keypoints_mean"] = round(float(np.mean(o_counts[:, 0])), 1)
    if not tile_override and not family:
        props = torch.cuda.get_device_properties(dev)
        vv = valu_view(name, pps, props.multi_processor_count, props.clock_rate * 1e3 if getattr(props, "clock_rate", 0) else 2.4e9)
        if vv and "valu_issue_frac" in vv:
            out["valu_issue_frac"] = vv["valu_issue_frac"]
            out["valu_issue_frac_all_4clk"] = vv["valu_issue_frac_all_4clk"]
            out["valu_wave_instr_per_pair"] = vv["valu_wave_instr_per_pair"]
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            kname = "k_compact_flat" if dom == "k_compact" else dom
            out["traffic"] = tj.get(name, {}).get(kname)
            out["traffic_note"] = "bytes per launch of %d %s, builder-measured rocprofv3 PMC pass (profiles/hbm_traffic.json), NOT measured in this run" % (
                P, "pairs" if dom in ("k_stereo", "k_median") else "images")
        except Exception:
            pass
    a.close(); b.close()
    return out


def measure_host_streamed(orb, torch, cfg, left_u, right_u, dev, P=256, seconds=1.5):
    """north-star regime (i): images in pinned host memory, one hipMemcpyAsync per LANE of the batch on the copy stream into a landing
    buffer that is read in place as level 0 (double buffered; a lane starts when its images have landed and its part of the buffer
    is refilled when the lane that used it two batches ago has finished), then the same kernels."""
    H, W, L, tile, th, fx, bf = cfg
    n_u = left_u.shape[0]
    idx = np.arange(P) % n_u
    lh = torch.from_numpy(left_u[idx]).pin_memory()
    rh = torch.from_numpy(right_u[idx]).pin_memory()
    bl = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, device_id=dev.index or 0, max_batch=P)
    br = orb.ORBExtractor(H, W, 1.2, L, 9, 14, 7, th, None, tile, tile, device_id=dev.index or 0, max_batch=P)
    ln, rn = lh.numpy(), rh.numpy()

    def step():
        bl.extract_batch_host_async(ln)
        br.extract_batch_host_async(rn)
        orb.stereo_match_batch_async(bl, br, bf / fx, bf)
    for _ in range(4):
        step()
    bl.sync(); br.sync()
    # a streaming run: the batches are enqueued back to back (upload of batch k+1 under the kernels of batch k) and the host waits once
    # at the end; the number of batches is calibrated from a short run so that the timed region lasts about `seconds`
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    bl.sync(); br.sync()
    n = max(16, int(8 * seconds / (time.perf_counter() - t0)))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    bl.sync(); br.sync()
    dt = time.perf_counter() - t0
    bl.close(); br.close()
    return {"value": round(n * P / dt, 1), "unit": "stereo pairs/s", "pcie_gb_per_s": round(n * P * 2 * H * W / dt / 1e9, 2),
            "sample": "%d steps of %d pairs from pinned host memory (jsorb_extract_batch_host_async), %.2f s" % (n, P, dt)}


def measure_frame_latency(cfg, left, right, frames=300, launches=3):
    """north-star regime: ONE stereo pair per call through the reference-shaped synchronous C++ API (two std::threads for L/R,
    SyncedMem::to_cpu x 4, ComputeStereoMatches) - tools/micro/frame_latency.cpp, built by __graft_entry__.build().  left / right: a few
    different pairs [n, H, W] that the driver rotates through (consecutive frames differ, as they do in a SLAM session)."""
    exe = os.path.join(ROOT, "tools", "micro", "frame_latency")
    if not os.path.exists(exe):
        return None
    import tempfile
    H, W, L, tile, th, fx, bf = cfg
    with tempfile.TemporaryDirectory() as td:
        lp, rp = os.path.join(td, "l.raw"), os.path.join(td, "r.raw")
        left = left if left.ndim == 3 else left[None]; right = right if right.ndim == 3 else right[None]
        left.tofile(lp); right.tofile(rp)
        env = dict(os.environ, JSORB_JSON="1", JSORB_ROTATE_PAIRS=str(left.shape[0]))
        try:
            # THREE process launches of the default shape (round-5 review: one launch per figure, 16 %% apart between boxes and runs): the figures of the
            # line are the medians over the launches, every launch's own median / p10 / p90 stays in `launches`
            runs = []
            for _ in range(launches):
                out = subprocess.run([exe, str(H), str(W), str(L), str(tile), str(th), str(fx), str(bf), lp, rp, str(frames)], env=env,
                                     capture_output=True, text=True, timeout=120)
                runs.append(json.loads(out.stdout.strip().splitlines()[-1]))
            res = dict(sorted(runs, key=lambda r_: r_["total_us_median"])[len(runs) // 2])      # the launch with the median median
            for key in ("total_us_median", "total_us_p10", "total_us_p90", "thread_spawn_us"):
                res[key] = round(median([r_[key] for r_ in runs]), 1)
            res["launches"] = [{k_: r_[k_] for k_ in ("total_us_median", "total_us_p10", "total_us_p90", "thread_spawn_us")} for r_ in runs]
            res["total_us_median_spread"] = [min(r_["total_us_median"] for r_ in runs), max(r_["total_us_median"] for r_ in runs)]
            res["threads"] = ("the driver process is bound to the cores of the GPU's NUMA node (jetson_slam_amd/placement.py; `placement` in this line) - its two "
                              "extractor threads are spawned per frame as in Frame.cpp:107-110; thread_spawn_us = two NO-OP std::threads spawned and joined, "
                              "timed in the same loop")
            res["what"] = ("C++ driver shaped like Frame::Frame: extract L||R in two std::threads + 4 x SyncedMem::to_cpu + ComputeStereoMatches, host images in "
                           "pageable memory; the match of frame k is enqueued by the library behind the extracts of frame k (jsorb_set_speculative_stereo) and "
                           "adopted by ComputeStereoMatches - total_us_plain is the same driver with JSORB_SPECULATE=0")
            out = subprocess.run([exe, str(H), str(W), str(L), str(tile), str(th), str(fx), str(bf), lp, rp, str(frames)], env=dict(env, JSORB_SPECULATE="0"),
                                 capture_output=True, text=True, timeout=120)
            plain = json.loads(out.stdout.strip().splitlines()[-1])
            res["total_us_plain"], res["total_us_median_plain"] = plain["total_us"], plain["total_us_median"]
            # not the reference's code shape: the same frame with the two extractor threads kept alive instead of spawned per frame
            out = subprocess.run([exe, str(H), str(W), str(L), str(tile), str(th), str(fx), str(bf), lp, rp, str(frames)], env=dict(env, JSORB_PERSISTENT_THREADS="1"),
                                 capture_output=True, text=True, timeout=120)
            res["total_us_median_persistent_threads"] = json.loads(out.stdout.strip().splitlines()[-1])["total_us_median"]
            # the shipped Frame's shape: four SyncedMem members constructed and destroyed with every frame (Frame.h:234-237)
            out = subprocess.run([exe, str(H), str(W), str(L), str(tile), str(th), str(fx), str(bf), lp, rp, str(frames)], env=dict(env, JSORB_FRESH_SYNCEDMEM="1"),
                                 capture_output=True, text=True, timeout=120)
            res["total_us_median_fresh_syncedmem_per_frame"] = json.loads(out.stdout.strip().splitlines()[-1])["total_us_median"]
            return res
        except Exception as e:      # never let a side measurement break the contract line
            return {"error": str(e)[:200]}


if __name__ == "__main__":
    main()
