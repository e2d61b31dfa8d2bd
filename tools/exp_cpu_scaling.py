"""Developer tool (run on the GPU box): why does the CPU oracle scale so poorly with processes there?  Prints the cgroup
quota / affinity / load and the oracle pair rate for several pool sizes."""
import os, sys, time
import multiprocessing as mp
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

print("usable:", bench._usable_cores(), "model:", bench._cpu_model(), "loadavg:", open("/proc/loadavg").read().strip())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(p, "=", open(p).read().strip().replace("\n", " | "))
    except Exception as e:
        print(p, "unreadable", e)
rng = np.random.default_rng(0)
t = (0.1 * rng.standard_normal(192000)).astype(np.float32)
bench._CPU_ITEMS = [((t + 0.01 * rng.standard_normal(192000)).astype(np.float32), t) for _ in range(64)]
bench._CPU_FN = bench.Cfg2.cpu_unit
import torch
torch.set_num_threads(1)
bench._cpu_run([0])
dt, _ = bench._cpu_run(list(range(16)))
print("1 thread: %.1f pairs/s" % (16 / dt))
for procs in (4, 16, 64, 128, 256):
    if procs > (os.cpu_count() or 1):
        break
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(bench._cpu_run, [[i % 64] for i in range(procs)], chunksize=1)
        best = 0
        for _ in range(3):
            t0 = time.perf_counter()
            r = pool.map(bench._cpu_run, [[(i * 2) % 64, (i * 2 + 1) % 64] for i in range(procs)], chunksize=1)
            wall = time.perf_counter() - t0
            best = max(best, 2 * procs / wall)
        inner = np.mean([x[0] for x in r]) / 2
        print("%3d procs: %.1f pairs/s (wall), mean in-worker time per pair %.1f ms" % (procs, best, inner * 1e3))
