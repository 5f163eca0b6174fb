"""bench.py's multi-GPU entry point.  CPU: argument / launcher consistency fails loudly before anything touches a GPU.
-m gpu: `python bench.py --gpus 2` with NO launcher starts two ranks itself; on the 1-GPU test box both ranks share cuda:0 and the
collective runs over gloo (JSORB_BENCH_SINGLE_DEVICE / JSORB_BENCH_BACKEND test hooks) - the product's N > 1 path (pair sharding,
jsorb_gather_counts_async, jsorb_stream_wait_done, all_gather, double-buffered payload) runs for real and is checked against the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_gpus_flag_must_match_the_launcher_world_size():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_clean_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_launches_two_ranks_itself_and_gathers_counts():
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "6", "--config", "c1", "--min-time", "0.05", "--no-cpu-baseline", "--no-extras",
            "--profile-steps", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(JSORB_BENCH_SINGLE_DEVICE="1", JSORB_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["pairs_per_step_total"] == 12
    assert d["parity_vs_oracle"] is True and d["gathered_counts_ok"] is True and d["parity_pairs_checked"] == 12
    # what makes a first real multi-GPU run diagnosable: every rank's own median block time next to the max the value is made of, and the time of the all_gather itself
    diag = d["multi_gpu_diag"]
    assert len(diag["per_rank"]) == 2 and all(r_["median_block_ms"] > 0 and r_["ms_per_step"] > 0 and r_["all_gather_ms_mean"] >= 0 and r_["all_gather_ms_max"] >= r_["all_gather_ms_mean"]
                                              for r_ in diag["per_rank"])
    assert max(r_["ms_per_step"] for r_ in diag["per_rank"]) <= d["ms_per_step"] * 1.5 and d["summary"]["pairs_per_s"] == d["value"]
    # round 6: the record checks itself (device / PCI / collective checks are "skipped" under the single-device gloo hook, never "failed")
    sc = diag["self_check"]
    assert set(sc["checks"]) == {"distinct_devices", "distinct_pci", "numa_bound", "all_gather_under_100us", "no_straggler_over_5pct"} and isinstance(sc["diagnosis"], str)
    assert isinstance(sc["checks"]["distinct_devices"], str) and isinstance(sc["checks"]["all_gather_under_100us"], str)
    assert all("pci" in r_ and "device_index" in r_ and "numa_bound" in r_ for r_ in diag["per_rank"])
    # --dry-multi: the self-check alone, in seconds; the exit code is non-zero iff a check failed (e.g. the two ranks of this box drift more than 5 % apart;
    # a rank exits with 3, which the launcher reports as a failed worker)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-multi", "--steps", "3", "--warmup", "1", "--pairs", "6", "--config", "c1"],
                       env=_clean_env(JSORB_BENCH_SINGLE_DEVICE="1", JSORB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    dd = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert dd["multi_gpu_diag"]["self_check"]["ok"] == (r.returncode == 0), r.stderr[-2000:]
    assert dd["cpu_baseline"] is None
    # strong scaling shape (BASELINE C4 style): the total is split over the ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--pairs-total", "8"],
                       env=_clean_env(JSORB_BENCH_SINGLE_DEVICE="1", JSORB_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["pairs_per_gpu_per_step"] == 4 and d["parity_vs_oracle"] is True


@pytest.mark.gpu
def test_bench_eight_ranks_c4_shape_on_one_device():
    """BASELINE C4 exactly as the driver's 8-GPU run shapes it - `--gpus 8 --pairs-total 64` on the EuRoC configuration: eight ranks (all
    on cuda:0 here, collective over gloo) of eight pairs each, global pair sharding, one all_gather per step, every rank checking every
    row of the gathered table, launch and teardown of eight processes."""
    args = ["--gpus", "8", "--pairs-total", "64", "--config", "c2", "--steps", "3", "--warmup", "1", "--min-time", "0.05", "--no-cpu-baseline", "--no-extras",
            "--profile-steps", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_clean_env(JSORB_BENCH_SINGLE_DEVICE="1", JSORB_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["pairs_per_gpu_per_step"] == 8 and d["config"]["pairs_per_step_total"] == 64
    assert d["parity_vs_oracle"] is True and d["gathered_counts_ok"] is True and d["parity_pairs_checked"] == 64


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)], env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "visible" in (r.stderr + r.stdout)
