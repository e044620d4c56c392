#!/usr/bin/env python3
"""Parity margin of the kernels' per-lane arithmetic WITHOUT a GPU: the host build of csrc/rmav_math.hpp (tests/hostmath, the very
same source the kernels compile) vs the fp64 oracle on N random single-step cases per kind (half from the reset distribution
U(-1,1), half 3x wider; actions over the Box) + every golden step case.  Prints the worst scaled error |d| / max(1, |y_ref|) per
state component group, split by tether branch for the slung-load kinds.  Used to choose the fp32 / fp64 split of the
slung-load integrators (bar: 1e-6).  Test infrastructure (uses oracle/).   python tools/parity_margin_host.py [N] [kinds...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("reinmav-gym_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np

import oracle as O
from gym_reinmav_amd import _abi as A
from util import KINDS, near_threshold, random_cases, scaled_err

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
kinds = sys.argv[2:] or list(KINDS)
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostmath")], check=True)
hm = C.CDLL(os.path.join(ROOT, "tests", "hostmath", "_build", "libhostmath.so"))
FP = C.POINTER(C.c_float)
GROUPS = {"quad3d": {"pos": (0, 3), "att": (3, 7), "vel": (7, 10)}, "quad3d_sl": {"pos": (0, 3), "att": (3, 7), "vel": (7, 10), "load_pos": (10, 13), "load_vel": (13, 16)},
          "quad2d": {"pos": (0, 2), "att": (2, 3), "vel": (3, 5)}, "quad2d_sl": {"pos": (0, 2), "att": (2, 3), "vel": (3, 5), "load_pos": (5, 7), "load_vel": (7, 9)}}
print("| kind | branch | cases | " + " | ".join(["worst"] + ["reward", "done mismatches away from a limit"]) + " | per group |")
print("|---|---|---|---|---|---|---|")
for kind in kinds:
    k = A.KIND_BY_NAME[kind]
    p = A.default_params(k, None)
    s, a = random_cases(kind, N, seed=77)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{kind}.npz"))
    s = np.concatenate([s, g["step_s"].astype(np.float32)])
    a = np.concatenate([a, g["step_a"].astype(np.float32)])
    n = len(s)
    s2 = s.copy()
    dist, done = np.zeros(n, np.float32), np.zeros(n, np.int32)
    assert hm.hm_step(k, C.byref(p), C.c_int64(n), s2.ctypes.data_as(FP), a.ctypes.data_as(FP), dist.ctypes.data_as(FP),
                      done.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    o2, r, d, _ = O.batch_step(kind, s.astype(np.float64), a.astype(np.float64))
    slack = O.tether_slack(kind, s.astype(np.float64))
    branches = {"all": np.ones(n, bool)} if np.isnan(slack).all() else {"taut": slack >= 1e-5, "slack": slack <= -1e-5}
    ok = near_threshold(kind, o2)
    for bn, m in branches.items():
        e = scaled_err(s2[m], o2[m])
        alive = m & ~d & ~done.astype(bool)
        re = scaled_err(-dist[alive], r[alive]).max()
        bad = int(((done.astype(bool) != d) & ~ok & m).sum())
        per = ", ".join(f"{gn} {e[:, lo:hi].max():.2e}" for gn, (lo, hi) in GROUPS[kind].items())
        print(f"| {kind} | {bn} | {int(m.sum())} | {e.max():.2e} | {re:.2e} | {bad} | {per} |", flush=True)
