"""world_size=2 `gloo` test of the multi-GPU batch path on CPU: pairs are sharded over ranks, each rank processes only its
own shard (here with the CPU oracle standing in for the GPU, which this container does not have) and the per-pair counts
are all-gathered - the one collective of the path.  The gathered table must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PAIRS, H, W = 5, 120, 160


def _counts_for(indices):
    sys.path.insert(0, ROOT)
    from jetson_slam_amd.synth import synth_stereo_pair
    from oracle import pyoracle as po
    kw = dict(height=H, width=W, n_levels=3, tile_h=12, tile_w=12)
    ol, orr = po.OracleExtractor(**kw), po.OracleExtractor(**kw)
    rows = []
    for i in indices:
        l, r = synth_stereo_pair(100 + i, H, W)
        ol.extract(l); orr.extract(r)
        _, _, st = po.stereo_match(ol, orr, 0.1, 20.0)
        rows.append([ol.n, orr.n, st["n_final"]])
    return torch.tensor(rows, dtype=torch.int32).reshape(-1, 3)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from jetson_slam_amd.batch import shard_range, all_gather_counts
    a, b = shard_range(N_PAIRS, rank, world)
    local = _counts_for(range(a, b))
    table = all_gather_counts(local, N_PAIRS)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), table.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_counts_gather_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = _counts_for(range(N_PAIRS)).numpy()
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.shape == (N_PAIRS, 3) and np.array_equal(got, ref)
    assert (ref[:, 0] > 20).all()
