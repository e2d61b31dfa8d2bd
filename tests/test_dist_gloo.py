"""CPU tests: the N>1 exchange (ssr_eval_amd.dist) with the gloo backend, world_size 2."""
import json
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datetime import datetime
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    from ssr_eval_amd import dist as D
    D.init_from_env(backend="gloo")
    assert D.rank_world() == (rank, world)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "aggregate.json")))
    work = [(s, f) for s in g["speakers"] for f in g["files"][s]]
    mine = D.shard_indices(len(work))
    assert list(mine) == list(range(rank, len(work), world))
    local = [g["per_file"][os.path.join(*work[i])] for i in mine]
    h = SSR_Eval_Helper(BasicTestee(), 44100, 44100, test_data_root=None)
    os.chdir(out_dir)
    final = h._assemble(work, g["speakers"], mine, local, rank == 0, datetime(2022, 1, 1))
    ok = final["averaged"] == g["averaged"] and final["each_speaker"] == g["each_speaker"]
    ok = ok and all(final[s][f] == g["per_file"][os.path.join(s, f)] for s, f in work)
    want = np.array([g["averaged"][k][m] for k in local[0] for m in ("lsd", "log_sispec", "sispec", "ssim")])
    ok = ok and np.allclose(h.last_allreduce_average, want, rtol=1e-13)
    # raw primitives
    red = D.allreduce_sums(np.full(5, rank + 1.0))
    ok = ok and np.array_equal(red, np.full(5, 3.0))
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("1" if ok else "0")
    import torch.distributed as dist
    dist.destroy_process_group()


def test_sharded_assembly_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "1" and open(tmp_path / "ok1").read() == "1"


def _worker_one_file(rank, world, port, out_dir):
    """world_size 2, ONE (speaker, file) item: rank 1 owns nothing (the idle rank used to die in a reshape
    while rank 0 blocked in the collective)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from datetime import datetime
    from ssr_eval_amd import SSR_Eval_Helper, BasicTestee
    from ssr_eval_amd import dist as D
    D.init_from_env(backend="gloo")
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "aggregate.json")))
    spk = g["speakers"][0]
    work = [(spk, g["files"][spk][0])]
    mine = D.shard_indices(len(work))
    assert len(mine) == (1 if rank == 0 else 0)
    local = [g["per_file"][os.path.join(*work[i])] for i in mine]
    h = SSR_Eval_Helper(BasicTestee(), 44100, 44100, test_data_root=None)
    os.chdir(out_dir)
    final = h._assemble(work, [spk], mine, local, False, datetime(2022, 1, 1))
    want = g["per_file"][os.path.join(*work[0])]
    ok = final[spk][work[0][1]] == want and final["averaged"] == want and final["each_speaker"][spk] == want
    ok = ok and list(final["averaged"].keys()) == list(want.keys())          # key order agreed on across ranks
    open(os.path.join(out_dir, "one%d" % rank), "w").write("1" if ok else "0")
    import torch.distributed as dist
    dist.destroy_process_group()


def test_sharded_assembly_with_an_idle_rank(tmp_path):
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_one_file, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "one0").read() == "1" and open(tmp_path / "one1").read() == "1"


def test_single_process_primitives():
    sys.path.insert(0, ROOT)
    from ssr_eval_amd import dist as D
    rows = np.arange(12.0).reshape(4, 3)
    out = D.allgather_rows(rows[[0, 2]], [0, 2], 4)
    assert np.array_equal(out[[0, 2]], rows[[0, 2]]) and np.isnan(out[1]).all()
    buf = D.speaker_sums(rows, [0, 0, 1, 1], 3)
    means, avg = D.mean_of_speaker_means(buf)
    assert means.shape == (2, 3) and np.allclose(avg, rows.reshape(2, 2, 3).mean(1).mean(0))


def _worker_width_mismatch(rank, world, port, out_dir):
    """Both ranks own rows but disagree on their width: every rank must raise (the narrower rank's rows used
    to be dropped silently and came back as NaN aggregates)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ssr_eval_amd import dist as D
    D.init_from_env(backend="gloo")
    rows = np.ones((2, 4 if rank == 0 else 8))
    try:
        D.allgather_rows(rows, [2 * rank, 2 * rank + 1], 4)
        got = "no error"
    except ValueError as e:
        got = "ValueError" if "disagree" in str(e) else repr(e)
    # a zero-width shard that owns rows is a mismatch too, not an idle rank
    try:
        D.allgather_rows(np.empty((1, 0)) if rank == 0 else np.ones((1, 4)), [rank], 2)
        got2 = "no error"
    except ValueError:
        got2 = "ValueError"
    open(os.path.join(out_dir, "w%d" % rank), "w").write(got + "," + got2)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_allgather_rows_width_mismatch_raises_on_every_rank(tmp_path):
    port = 33500 + os.getpid() % 2000
    mp.spawn(_worker_width_mismatch, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "w0").read() == "ValueError,ValueError" and open(tmp_path / "w1").read() == "ValueError,ValueError"


def _worker_one_collective(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ssr_eval_amd import dist as D
    D.init_from_env(backend="gloo")
    rng = np.random.default_rng(5)
    n, K, S = 37, 6, 4
    rows = rng.standard_normal((n, K)) * 10.0 ** rng.integers(-3, 4, (n, 1))
    spk = rng.integers(0, S, n)
    lens = rng.integers(1000, 90000, n)
    mine = D.shard_indices_balanced(lens)                       # uneven counts per rank: the padded capacity is agreed on
    import torch.distributed as dist
    calls = {"n": 0}
    orig = dist.all_gather

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    dist.all_gather = counting
    table, buf = D.gather_rows_and_speaker_sums(rows[mine], mine, n, spk[mine], S)
    dist.all_gather = orig
    ok = calls["n"] == 1 and np.array_equal(table, rows)
    want = np.zeros((S, K + 1))
    for r in range(world):                                       # the documented order: rank blocks added in rank order
        want += D.speaker_sums(rows[D.shard_indices_balanced(lens, r, world)], spk[D.shard_indices_balanced(lens, r, world)], S)
    ok = ok and np.array_equal(buf, want)
    np.save(os.path.join(out_dir, "buf%d.npy" % rank), buf)
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


def test_rows_and_speaker_sums_in_one_collective_world3(tmp_path):
    """the per-file rows and the per-speaker sums + counts cross in ONE all-gather; every rank adds the speaker
    blocks in rank order, so the aggregate buffer is BIT-identical on every rank (length-balanced, uneven shards)."""
    port = 29500 + (os.getpid() + 911) % 2000
    mp.spawn(_worker_one_collective, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(open(tmp_path / ("ok%d" % r)).read() == "1" for r in range(3))
    b = [np.load(tmp_path / ("buf%d.npy" % r)) for r in range(3)]
    assert b[0].tobytes() == b[1].tobytes() == b[2].tobytes()


def test_decode_threads_share_the_granted_cores(monkeypatch):
    """the reader pool of a process is its share of the cores the JOB may use (scheduler affinity / cgroup quota
    divided by the ranks on this node), not min(16, os.cpu_count()) per process."""
    sys.path.insert(0, ROOT)
    from ssr_eval_amd import io as sio
    monkeypatch.delenv("SSR_DECODE_THREADS", raising=False)
    monkeypatch.setattr(sio, "usable_cores", lambda: 16)
    for lws, want in ((1, 16), (2, 8), (8, 2), (16, 1), (64, 1)):
        monkeypatch.setenv("LOCAL_WORLD_SIZE", str(lws))
        assert sio.decode_threads() == want
    monkeypatch.setattr(sio, "usable_cores", lambda: 256)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert sio.decode_threads() == 16                              # the cap
    monkeypatch.setenv("SSR_DECODE_THREADS", "3")
    assert sio.decode_threads() == 3
    monkeypatch.undo()
    assert 1 <= sio.usable_cores() <= len(os.sched_getaffinity(0))


def _worker_decode_tree(rank, world, port, root, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    os.environ.pop("SSR_DECODE_THREADS", None)
    import threading
    from ssr_eval_amd import dist as D
    from ssr_eval_amd import io as sio
    D.init_from_env(backend="gloo")
    files = sorted(os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith(".wav"))
    mine = D.shard_indices(len(files))
    got = sio.decode_batch([files[i] for i in mine])
    wait = sio.decode_async([files[i] for i in mine])
    got2 = wait()
    readers = sum(1 for t in threading.enumerate() if t.name.startswith("ssr-decode"))
    sums = np.array([[float(np.abs(w).sum()), float(sr)] for (w, sr) in got]).reshape(-1, 2)
    same = all(np.array_equal(a[0], b[0]) for a, b in zip(got, got2))
    table = D.allgather_rows(sums, mine, len(files))
    json.dump({"readers": readers, "limit": sio.decode_threads(), "cores": sio.usable_cores(), "same": same,
               "table": table.tolist()}, open(os.path.join(out_dir, "dec%d.json" % rank), "w"))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_world_size_8_file_tree_reader_threads_within_the_affinity(tmp_path):
    """EIGHT gloo ranks decode their round-robin shards of one wav tree (the host stage of evaluate(): the kernels need a GPU,
    tests/test_gpu_configs.py runs the whole evaluate() under eight processes) - every rank's pool is its share of the cores, the
    job as a whole starts no more reader threads than max(ranks, usable cores), and the gathered table is the one-process one."""
    sys.path.insert(0, ROOT)
    from ssr_eval_amd.io import write_wav, read_audio, usable_cores
    rng = np.random.default_rng(3)
    root = tmp_path / "tree"
    for s, c in (("p1", 9), ("p2", 7), ("p3", 5)):
        (root / s).mkdir(parents=True)
        for i in range(c):
            write_wav(str(root / s / ("u%02d.wav" % i)), 0.1 * rng.standard_normal(int(rng.integers(2000, 9000))), 16000)
    port = 29500 + (os.getpid() + 1213) % 2000
    mp.spawn(_worker_decode_tree, args=(8, port, str(root), str(tmp_path)), nprocs=8, join=True)
    d = [json.load(open(tmp_path / ("dec%d.json" % r))) for r in range(8)]
    assert all(x["same"] for x in d)
    assert all(1 <= x["readers"] <= x["limit"] for x in d)
    assert sum(x["readers"] for x in d) <= max(8, usable_cores())
    files = sorted(os.path.join(dd, f) for dd, _, fs in os.walk(root) for f in fs if f.endswith(".wav"))
    want = [[float(np.abs(read_audio(f)[0]).sum()), 16000.0] for f in files]
    assert all(x["table"] == want for x in d)
