"""Developer experiment: engine clock and socket power while one stage of the pair pipeline runs back to back (sysfs, sampled
from a thread), against the idle readings.  STAGE = stft (default: LSD + magnitudes) | lsd (no magnitude stores) | ssim | full."""
import glob, os, sys, threading, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools')); import devlib; devlib.select()   # SSR_DEV_LIB: alternative build
import bench
from ssr_eval_amd import backend as B


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def sample():
    out = {}
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"):
        v = read(f)
        if v: out["sclk_MHz"] = round(int(v) / 1e6)
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + \
            glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        v = read(f)
        if v: out["power_W"] = round(int(v) / 1e6)
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        v = read(f)
        if v:
            act = [l for l in v.splitlines() if l.endswith("*")]
            out["dpm_sclk"] = act[0] if act else v.replace("\n", " | ")
    return out


def main():
    dev = torch.device("cuda", 0)
    n = 1024
    g = torch.Generator(device=dev).manual_seed(1)
    tgt = (0.1 * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    est = (tgt + 0.01 * torch.randn((n, bench.N_SAMPLES), generator=g, device=dev)).contiguous()
    b = B.PairBatch(B.get_plan(2048, 512, "f64", dev), B.Ragged.from_uniform(est), B.Ragged.from_uniform(tgt))
    mask, st = {"stft": (B.M_LSD | B.M_SSIM, 1), "lsd": (B.M_LSD, 1), "ssim": (B.M_SSIM, 2), "full": (B.M_LSD | B.M_SSIM, 7)}[os.environ.get("STAGE", "stft")]
    b.run(mask, stages=st); torch.cuda.synchronize()
    print(json.dumps({"idle": sample()}))
    seen, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            seen.append(sample()); time.sleep(0.05)
    th = threading.Thread(target=poll); th.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(50): b.run(mask, stages=st)
        torch.cuda.synchronize(); k += 50
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    print(json.dumps({"stage": os.environ.get("STAGE", "stft"), "ms_per_run": round(1e3 * dt / k, 4), "samples": seen[4::8]}))


if __name__ == "__main__":
    main()
