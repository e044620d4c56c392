#!/usr/bin/env python3
"""Summarises tools/throttle_probe.sh: per workload the change of every numeric field of `amd-smi metric --throttle` (residency accumulators),
the power / clock readings, and the launch-time curve (first slices vs the rest)."""
import glob, json, os, re, sys
d = sys.argv[1]


def flat(x, pre=""):
    out = {}
    if isinstance(x, dict):
        for k, v in x.items():
            out.update(flat(v, pre + k + "."))
    elif isinstance(x, list):
        for i, v in enumerate(x):
            out.update(flat(v, pre + f"{i}."))
    else:
        out[pre[:-1]] = x
    return out


def num(v):
    if isinstance(v, (int, float)):
        return float(v)
    if isinstance(v, dict) and "value" in v:
        return num(v["value"])
    try:
        return float(str(v).split()[0])
    except Exception:
        return None


def load(name):
    try:
        return flat(json.load(open(os.path.join(d, f"smi_{name}.json"))))
    except Exception as e:
        return {"error": repr(e)}


series = sorted(glob.glob(os.path.join(d, "series_*.txt")))
print("# Throttle residencies and launch-time curves (tools/throttle_probe.sh)\n")
for f in series:
    m = re.match(r"series_(\d+)_(\w+?)_(\d+)_(\w+)\.txt", os.path.basename(f))
    i, mode, n, kind = m.groups()
    L = [tuple(map(float, l.split()[1:])) for l in open(f) if l.startswith("L ")]
    S = [tuple(map(float, l.split()[1:])) for l in open(f) if l.startswith("S ")]
    early = [u for t, u in L if t < 0.05]
    late = [u for t, u in L if t > L[-1][0] / 2]
    print(f"## {i}. {mode} {kind} {n} envs")
    if L:
        print(f"- launch time: first 50 ms {min(early) if early else float('nan'):.1f} - {max(early) if early else float('nan'):.1f} us, second half {sum(late) / len(late):.1f} us "
              f"(min {min(late):.1f}, max {max(late):.1f}); curve: " + " ".join(f"{t:.2f}s:{u:.0f}" for t, u in L[:: max(1, len(L) // 24)]))
    if S:
        h = [s for s in S if s[0] > S[-1][0] / 2]
        print(f"- second half: power {sum(s[1] for s in h) / len(h):.0f} W, sclk {sum(s[2] for s in h) / len(h):.0f} MHz (min {min(s[2] for s in h):.0f}), "
              f"fclk {sum(s[3] for s in h) / len(h):.0f}, mclk {sum(s[4] for s in h) / len(h):.0f}; first 100 ms: power {max(s[1] for s in S if s[0] < 0.1) if any(s[0] < 0.1 for s in S) else float('nan'):.0f} W, "
              f"sclk {max(s[2] for s in S if s[0] < 0.1) if any(s[0] < 0.1 for s in S) else float('nan'):.0f} MHz")
    a, b = load(f"before_{i}"), load(f"after_{i}")
    moved = []
    for k in sorted(b):
        x, y = num(a.get(k)), num(b.get(k))
        if x is not None and y is not None and y != x and any(w in k.lower() for w in ("throttle", "residency", "accum", "violation", "prochot", "ppt", "thm", "thermal")):
            moved.append(f"{k}: {x:g} -> {y:g} (+{y - x:g})")
    print("- throttle fields that moved: " + ("; ".join(moved) if moved else "none"))
    keys = [k for k in b if any(w in k.lower() for w in ("socket_power", "hotspot", "temperature", "gfx_0.clk", "mem_0.clk", "clk"))][:14]
    print("- after: " + ", ".join(f"{k.split('.', 2)[-1]}={b[k]}" for k in keys))
    print()
