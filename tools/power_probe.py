#!/usr/bin/env python3
"""Sustained speed, socket power and shader clock of one workload (hwmon / sysfs sampled every 20 ms from a thread).
MODE = rollout | memset | compute (rollout with no trajectory outputs);  N, KIND, ACTIONS, SECS as environment variables."""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
n, T, R = int(os.environ.get("N", "131072")), 64, 6
mode, kind, actions = os.environ.get("MODE", "rollout"), os.environ.get("KIND", "quad3d"), os.environ.get("ACTIONS", "random")
secs = float(os.environ.get("SECS", "3"))
dev = torch.device("cuda", 0)
env = g.BatchedQuadrotor(kind, n, seed=0)
if os.environ.get("TUNE"):   # e.g. TUNE=split=0,store_policy=2 (rmav_set_tuning overrides)
    env.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["TUNE"].split(",")})
nS, nA = env.nS, env.nA
want = () if mode == "compute" else ("actions", "obs", "rew", "done")
ring = [dict(actions=torch.zeros((T, nA, n), device=dev), obs=torch.zeros((T, nS, n), device=dev),
             rew=torch.zeros((T, n), device=dev), done=torch.zeros((T, n), dtype=torch.uint8, device=dev)) for _ in range(R)]
nbytes = n * (T * (4 * (nS + nA + 1) + 1) + 8 * nS + 24)
flat = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(R)] if mode == "memset" else None
pr = torch.cuda.get_device_properties(0)
bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
devdir = f"/sys/bus/pci/devices/{bdf}"
hw = (sorted(glob.glob(devdir + "/hwmon/hwmon*")) or [None])[0]
f_pow = hw and (hw + "/power1_input" if os.path.exists(hw + "/power1_input") else hw + "/power1_average")
f_clk = hw and hw + "/freq1_input"
def cur_level(name):   # "1: 2100Mhz *" -> 2100
    try:
        for l in open(f"{devdir}/{name}").read().splitlines():
            if l.rstrip().endswith("*"):
                return float(l.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except Exception:
        pass
    return float("nan")
if os.environ.get("SHOW_CAP") == "1" and hw:
    for f in ("power1_cap", "power1_cap_default", "power1_cap_max", "power1_cap_min"):
        try: print(f, int(open(hw + "/" + f).read()) / 1e6, "W")
        except Exception as e: print(f, e)
    for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        try: print(f, open(devdir + "/" + f).read().replace("\n", " | "))
        except Exception as e: print(f, e)
samples, stop = [], False
def sampler():
    while not stop:
        try:
            p = int(open(f_pow).read()) / 1e6 if f_pow else float("nan")
            c = int(open(f_clk).read()) / 1e6 if f_clk else float("nan")
            samples.append((time.perf_counter(), p, c, cur_level("pp_dpm_fclk"), cur_level("pp_dpm_mclk")))
        except Exception:
            pass
        time.sleep(0.02)
th = threading.Thread(target=sampler, daemon=True); th.start()
torch.cuda.synchronize()
t_start = time.perf_counter(); rows, i = [], 0
while time.perf_counter() - t_start < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        if mode == "memset":
            flat[i % R].zero_()
        else:
            env.rollout(T, mode=actions, want=want, device_out=True, out=ring[i % R] if want else None)
        i += 1
    e1.record(); torch.cuda.synchronize()
    rows.append((time.perf_counter() - t_start, e0.elapsed_time(e1) * 10))
stop = True; th.join()
if os.environ.get("SERIES"):   # time series: launch time per 100-launch slice, and the sampled power / clocks
    with open(os.environ["SERIES"], "w") as f:
        f.write("# t_s us_per_launch\n" + "".join(f"L {t:.3f} {u:.2f}\n" for t, u in rows))
        f.write("# t_s power_w sclk_mhz fclk_mhz mclk_mhz\n" + "".join(f"S {t - t_start:.3f} {p:.0f} {c:.0f} {fc:.0f} {m:.0f}\n" for t, p, c, fc, m in samples))
steady = [u for t, u in rows if t > secs / 2]
burst = min(u for t, u in rows)
sp = [(p, c, f, m) for t, p, c, f, m in samples if t - t_start > secs / 2]
avg = lambda xs: sum(xs) / max(1, len(xs))
print(f"{mode:8s} {kind} {actions} n={n} tune={os.environ.get('TUNE', '')} lib={os.path.basename(os.environ.get('RMAV_LIB_PATH', 'default'))}: "
      f"best slice {burst:6.1f} us, steady {avg(steady):6.1f} us ({nbytes / avg(steady) / 1e6:.2f} TB/s), "
      f"power {avg([x[0] for x in sp]):5.0f} W, sclk {avg([x[1] for x in sp]):5.0f} (min {min([x[1] for x in sp] or [0]):.0f}) fclk {avg([x[2] for x in sp]):5.0f} mclk {avg([x[3] for x in sp]):5.0f} MHz, {len(sp)} samples [{bdf}]", flush=True)
