"""Utterance-level data parallelism: one process per GPU, RCCL over xGMI through torch.distributed.

The path shards naturally (every (utterance, degradation) pair is independent, ssr_eval/eval.py:136-154,
193-198); the only exchange is the aggregation of ssr_eval/eval.py:200-216:

* ``allreduce_sums``  - ONE float64 SUM all-reduce of the per-speaker [sums..., count] buffer (a few
  hundred bytes; latency-bound), from which every rank forms the mean of per-speaker means;
* ``allgather_rows``  - ONE padded all-gather of the per-utterance metric rows (for the per-file JSON
  block and for a bit-identical np.mean in the reference's order).

Backend "nccl" IS RCCL on ROCm; CPU tests use "gloo" with world_size 2.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (no-op for single-process runs)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, init_method="env://")


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n, rank=None, world=None):
    """Static round-robin ownership of global item indices."""
    if rank is None:
        rank, world = rank_world()
    return np.arange(rank, n, world)


def _comm_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_sums(buf):
    """In-place float64 SUM all-reduce of a numpy array / tensor; returns a numpy array."""
    t = torch.as_tensor(np.asarray(buf, dtype=np.float64)) if not isinstance(buf, torch.Tensor) else buf.double()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = t.to(_comm_device()).contiguous()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allgather_rows(local_rows, global_index, n_total):
    """Scatter rank-local rows [n_local, K] (owning global indices `global_index`) into [n_total, K] on every rank."""
    local_rows = np.asarray(local_rows, dtype=np.float64)
    # a rank that owns nothing passes an empty array: its width is whatever the other ranks agree on (MAX below)
    local_rows = local_rows.reshape(len(global_index), -1) if local_rows.size else np.empty((len(global_index), 0))
    K = local_rows.shape[1]
    rank, world = rank_world()
    if world == 1:
        out = np.full((n_total, K), np.nan)
        out[np.asarray(global_index, dtype=np.int64)] = local_rows
        return out
    # Width agreement: only a rank that owns NO row takes its width from the others.  Ranks that own rows must agree - a
    # key / metric mismatch between shards is an error on EVERY rank (all of them see the same MAX / MIN), never NaN rows.
    owns = len(global_index) > 0
    kk = torch.tensor([K if owns else -1, -K if owns else -(1 << 60)], dtype=torch.int64, device=_comm_device())
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    k_max, k_min = int(kk[0].item()), -int(kk[1].item())
    if k_max >= 0 and k_min != k_max:
        raise ValueError("allgather_rows: ranks disagree on the row width (%d .. %d columns): the shards were evaluated with "
                         "different keys or metrics" % (k_min, k_max))
    K = max(k_max, 0)
    cap = -(-n_total // world)
    pack = torch.full((cap, K + 1), float("nan"), dtype=torch.float64)
    pack[:, 0] = -1.0
    if owns:
        pack[:len(global_index), 0] = torch.as_tensor(np.asarray(global_index, dtype=np.float64))
        pack[:len(global_index), 1:] = torch.as_tensor(local_rows)
    pack = pack.to(_comm_device())
    gathered = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(gathered, pack)
    out = np.full((n_total, K), np.nan)
    for g in gathered:
        g = g.cpu().numpy()
        ok = g[:, 0] >= 0
        out[g[ok, 0].astype(np.int64)] = g[ok, 1:]
    return out


def speaker_sums(rows, speaker_ids, n_speakers):
    """[n_speakers, K + 1]: per-speaker column sums of `rows` plus the row count in the last column."""
    rows = np.asarray(rows, dtype=np.float64)
    K = rows.shape[1] if rows.ndim == 2 else 0     # an empty shard still carries its width: shape (0, K)
    buf = np.zeros((n_speakers, K + 1))
    for r, s in zip(rows, speaker_ids):
        buf[s, :K] += r
        buf[s, K] += 1.0
    return buf


def mean_of_speaker_means(buf):
    """(per-speaker means [S, K], mean over speakers [K]) from an all-reduced speaker_sums buffer
    (ssr_eval/eval.py:200-216: the published aggregate is a mean of per-speaker means, not a global mean)."""
    K = buf.shape[1] - 1
    present = buf[:, K] > 0
    means = buf[present, :K] / buf[present, K:K + 1]
    return means, means.mean(axis=0)
