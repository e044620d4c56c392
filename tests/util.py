"""Shared helpers for the parity tests."""
import numpy as np

KINDS = ("quad2d", "quad2d_sl", "quad3d", "quad3d_sl")
NS = {"quad2d": 5, "quad2d_sl": 9, "quad3d": 10, "quad3d_sl": 16}
NA = {"quad2d": 2, "quad2d_sl": 2, "quad3d": 4, "quad3d_sl": 4}
BOX = {"quad2d": (-10.0, 10.0), "quad2d_sl": (-10.0, 10.0), "quad3d": (0.0, 10.0), "quad3d_sl": (-10.0, 10.0)}
# (slice of the position that terminates, slice of the velocity that terminates, pos limit, vel limit)
TERM = {
    "quad2d": (slice(0, 2), slice(3, 5), 3.0, 2.0),
    "quad2d_sl": (slice(5, 7), slice(7, 9), 2.0, 10.0),
    "quad3d": (slice(0, 3), slice(7, 10), 3.0, 10.0),
    "quad3d_sl": (slice(10, 13), slice(7, 10), 3.0, 10.0),
}
# north_star tolerance: |d| <= 1e-6 * max(1, |y_ref|) on state and reward, fp32 path vs fp64 reference
TOL = 1e-6
# controllers: computed in fp64 on fp32 inputs, output rounded to fp32 -> same bar
CTRL_TOL = 1e-6


def scaled_err(x, ref):
    x, ref = np.asarray(x, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(x - ref) / np.maximum(1.0, np.abs(ref))


def near_threshold(kind, s_next, eps=1e-5, limits=None):
    """Envs whose terminating norms are within eps of a limit: either `done` answer is accepted."""
    ps, vs, pl, vl = TERM[kind]
    if limits is not None:
        pl, vl = limits
    s_next = np.asarray(s_next, dtype=np.float64)
    return (np.abs(np.linalg.norm(s_next[:, ps], axis=1) - pl) < eps) | (
        np.abs(np.linalg.norm(s_next[:, vs], axis=1) - vl) < eps)


def random_cases(kind, n, seed, wide=True):
    rng = np.random.RandomState(seed)
    lo, hi = BOX[kind]
    sc = np.where(np.arange(n) % 2 == 0, 1.0, 3.0 if wide else 1.0)[:, None]
    s = (rng.uniform(-1, 1, (n, NS[kind])) * sc).astype(np.float32)
    a = rng.uniform(lo, hi, (n, NA[kind])).astype(np.float32)
    return s, a
