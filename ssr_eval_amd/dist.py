"""Utterance-level data parallelism: one process per GPU, RCCL over xGMI through torch.distributed.

The path shards naturally (every (utterance, degradation) pair is independent, ssr_eval/eval.py:136-154,
193-198); the only exchange is the aggregation of ssr_eval/eval.py:200-216:

* ``allreduce_sums``  - ONE float64 SUM all-reduce of the per-speaker [sums..., count] buffer (a few
  hundred bytes; latency-bound), from which every rank forms the mean of per-speaker means;
* ``allgather_rows``  - ONE padded all-gather of the per-utterance metric rows (for the per-file JSON
  block and for a bit-identical np.mean in the reference's order);
* ``gather_rows_and_speaker_sums`` (round 4) - both in ONE all-gather (behind one 24-byte MAX all-reduce that agrees on width and
  largest shard): the speaker sums ride behind the rows and are added in rank order on every rank (at 8 GPUs a cfg-4 step is
  ~2 ms of kernels: every further blocking collective per step would show);
* ``shard_indices_balanced`` - length-balanced dealing (longest first to the least-loaded rank).

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


def shard_indices_balanced(lengths, rank=None, world=None):
    """Length-balanced ownership ("sort by n, deal greedily"): items in descending length, each to the rank that
    holds the fewest samples so far (ties: the lowest rank) - the longest-processing-time rule, deterministic, the same on every
    rank.  Returns this rank's global indices in ascending order.  On the 2,937-utterance VCTK-shaped set the heaviest shard is
    within 0.1 % of the mean for 2, 4 and 8 ranks (round-robin: 1-3 %), which matters once a step is ~2 ms of kernels."""
    if rank is None:
        rank, world = rank_world()
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.lexsort((np.arange(len(lengths)), -lengths))         # descending length, index as the tie-break
    load = np.zeros(world, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int64)
    import heapq
    heap = [(0, r) for r in range(world)]
    for i in order:
        l, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (l + int(lengths[i]), r))
    return np.nonzero(owner == rank)[0]


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


def gather_rows_and_speaker_sums(local_rows, global_index, n_total, speaker_ids, n_speakers):
    """The path's whole data exchange in ONE all-gather (ssr_eval/eval.py:200-216 needs the per-file rows for the JSON block and
    the per-speaker sums + counts for the aggregate), behind one 24-byte MAX all-reduce that agrees on the row width and the
    largest shard: every rank appends its [n_speakers, K + 1] speaker_sums rows to its padded per-utterance rows, the all-gather
    moves both, and every rank adds the gathered speaker blocks in rank order - a fixed order, so the aggregate is bit-identical
    on every rank (an all-reduce leaves the order to the library).  Two blocking collectives per evaluation in all (three in
    round 4; evaluate() adds one all_gather_object for the file names of the JSON block).
    -> (table [n_total, K], speaker buffer [n_speakers, K + 1])."""
    local_rows = np.asarray(local_rows, dtype=np.float64)
    local_rows = local_rows.reshape(len(global_index), -1) if local_rows.size else np.empty((len(global_index), 0))
    rank, world = rank_world()
    sums = speaker_sums(local_rows, speaker_ids, n_speakers)
    if world == 1:
        return allgather_rows(local_rows, global_index, n_total), sums
    owns = len(global_index) > 0
    K = local_rows.shape[1]
    # ONE 24-byte MAX all-reduce carries the width agreement (as in allgather_rows: a mismatch raises on EVERY rank) and the largest
    # shard (balanced shards may exceed ceil(n / world) by a few rows) - then ONE all-gather carries everything else
    kk = torch.tensor([K if owns else -1, -K if owns else -(1 << 60), len(global_index)], dtype=torch.int64, device=_comm_device())
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    k_max, k_min = int(kk[0].item()), -int(kk[1].item())
    if k_max >= 0 and k_min != k_max:
        raise ValueError("gather_rows_and_speaker_sums: ranks disagree on the row width (%d .. %d columns): the shards were "
                         "evaluated with different keys or metrics" % (k_min, k_max))
    K = max(k_max, 0)
    cap = max(-(-n_total // world), int(kk[2].item()))
    pack = torch.full((cap + n_speakers, K + 1), float("nan"), dtype=torch.float64)
    pack[:cap, 0] = -1.0
    if owns:
        pack[:len(global_index), 0] = torch.as_tensor(np.asarray(global_index, dtype=np.float64))
        pack[:len(global_index), 1:] = torch.as_tensor(local_rows)
    blk = np.zeros((n_speakers, K + 1))
    if owns:
        blk[:, :K] = sums[:, :K]
    blk[:, K] = sums[:, -1]
    pack[cap:] = torch.as_tensor(blk)
    pack = pack.to(_comm_device())
    gathered = [torch.empty_like(pack) for _ in range(world)]
    dist.all_gather(gathered, pack)
    out = np.full((n_total, K), np.nan)
    buf = np.zeros((n_speakers, K + 1))
    for g in gathered:                                               # rank order: the same float64 additions on every rank
        g = g.cpu().numpy()
        ok = g[:cap, 0] >= 0
        out[g[:cap][ok, 0].astype(np.int64)] = g[:cap][ok, 1:]
        buf += g[cap:]
    return out, buf


def allreduce_max_int(v):
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


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
