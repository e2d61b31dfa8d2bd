"""CPU tests of bench.py's launcher: --gpus N spawns N ranks by itself, times between barriers, takes the MAX over
ranks, all-reduces the per-step sums, and can never report fewer ranks than asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True,
                          timeout=300, env=e)


def test_self_launch_two_ranks_gloo():
    r = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--_cpu-skeleton")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # exactly ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["extra"]["job_means"] == [1.5]                           # (1 + 2) / (1 + 1): both ranks' sums arrived
    assert d["extra"]["allreduce_payload_bytes_per_step"] == 16
    assert abs(d["value"] - 10 * 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3   # whole-job units / max-rank time


def test_more_gpus_than_devices_fails_loudly():
    r = _bench("--gpus", "8", "--steps", "1", "--warmup", "0", env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and "--gpus 8" in (r.stderr + r.stdout) and "{" not in r.stdout


def test_world_size_mismatch_fails_loudly():
    r = _bench("--gpus", "2", "--_cpu-skeleton", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE is 1" in (r.stderr + r.stdout)


def test_torchrun_launch_two_ranks_gloo():
    """The driver's own launch line for N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--_cpu-skeleton"], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["extra"]["job_means"] == [1.5]


def test_cfg4_strong_scaling_shards_and_collectives_gloo():
    """bench.py --config cfg4 (the 2,937-utterance sharded test set): with N = 1, 2 and 3 ranks (gloo, kernels replaced by
    functions of the global utterance index) the [speakers, sums + count] blocks that ride behind the rows of the step's ONE
    all-gather yield the SAME mean of speaker means, the gather delivers every utterance's row, the shards are length-balanced,
    and the line says strong scaling with value = 2,937 x steps / time."""
    import numpy as np
    means = {}
    for n in (1, 2, 3):
        r = _bench("--config", "cfg4", "--gpus", str(n), "--steps", "2", "--warmup", "1", "--_cpu-skeleton")
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["scaling"] == "strong" and d["unit"] == "pairs/s"
        assert abs(d["value"] - 2937 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3       # NOT multiplied by the rank count
        assert d["extra"]["allgather_rows_received"] == 2937
        assert abs(d["extra"]["shard_utterances"] - 2937 / n) <= 8          # length-balanced dealing: nearly equal counts too
        assert d["extra"]["shard_balance_max_over_mean"] <= 1.01
        means[n] = np.array(d["extra"]["job_means"])
    # reference value: mean over speakers of per-speaker means of the four index functions
    counts = [424, 424, 123, 419, 301, 424, 424, 398]
    gi = np.arange(2937, dtype=np.float64)
    vals = np.stack([gi, 0.5 * gi, gi * gi * 1e-3, np.cos(gi)], axis=1)
    want = np.mean([vals[s:s + c].mean(axis=0) for s, c in zip(np.cumsum([0] + counts[:-1]), counts)], axis=0)
    for n in (1, 2, 3):
        np.testing.assert_allclose(means[n], want, rtol=1e-12, atol=1e-12)


def test_true_world_size_8_launcher_and_accounting_gloo():
    """The driver's 8-GPU launch, on CPU: eight gloo ranks through bench.py's own launcher for cfg-2, cfg-5 and cfg-4 - one JSON
    line from rank 0, n_gpus = the ranks that joined, MAX-over-ranks time, whole-job units: cfg-2 8 x 1024 pairs per step (weak),
    cfg-5 8 x 12,500 utterances = 100 k x 192,000 samples per step (weak), cfg-4 the 2,937-utterance set per step whatever N
    (strong), with either collective - BASELINE cfg-4 names the all-reduce, evaluate() needs the all-gather."""
    import numpy as np
    r = _bench("--config", "cfg2", "--gpus", "8", "--steps", "2", "--warmup", "1", "--pairs", "1000", "--_cpu-skeleton")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 1000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    assert d["extra"]["job_means"] == [4.5]                           # (1 + ... + 8) / 8: all eight ranks' sums arrived
    r = _bench("--config", "cfg5", "--gpus", "8", "--steps", "2", "--warmup", "1", "--_cpu-skeleton")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 8 and abs(d["value"] - 100000 * 192000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    means = {}
    for coll in ("allgather", "allreduce"):
        r = _bench("--config", "cfg4", "--gpus", "8", "--steps", "2", "--warmup", "1", "--collective", coll, "--_cpu-skeleton")
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["extra"]["collective"] == coll
        assert abs(d["value"] - 2937 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
        assert d["extra"]["shard_balance_max_over_mean"] <= 1.01 and abs(d["extra"]["shard_utterances"] - 2937 / 8) <= 8
        if coll == "allgather":
            assert d["extra"]["allgather_rows_received"] == 2937
        means[coll] = np.array(d["extra"]["job_means"])
    np.testing.assert_allclose(means["allreduce"], means["allgather"], rtol=1e-12, atol=1e-12)
