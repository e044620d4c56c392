#!/usr/bin/env python3
"""Container-only: times the REFERENCE's own step() (the files under /root/reference, loaded by oracle/ref_harness.py) beside the
build's CPU restatements on the same host core, and writes the ratios to profiles/cpu_calibration.json (BASELINE.md section 4,
item 1).  bench.py copies that file into its JSON line as `calibration`, so that the CPU numbers it measures on the GPU box's
host (where the reference cannot go) can be related to the reference:  reference-equivalent steps/s on the box
~= cpu_baseline.value / ratio.  Test infrastructure; nothing in the product reads it.

    python oracle/calibrate.py            # ~1 min, needs /root/reference
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy_ref   # noqa: E402
import oracle as O   # noqa: E402
import ref_harness as R   # noqa: E402

BOX = {"quad2d": (-10.0, 10.0), "quad2d_sl": (-10.0, 10.0), "quad3d": (0.0, 10.0), "quad3d_sl": (-10.0, 10.0)}


def time_reference(kind, seconds=4.0):
    """Random actions over the kind's Box, reset on done: the loop of BASELINE.md section 2."""
    env = R.RefEnv(kind, seed=0)
    env.reset()
    rng = np.random.RandomState(0)
    lo, hi = BOX[kind]
    acts = rng.uniform(lo, hi, (4096, O.ACTION_DIM[kind]))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for k in range(512):
            _, _, d = env.step(acts[(n + k) & 4095])
            if d:
                env.reset()
        n += 512
    return n / (time.perf_counter() - t0)


def time_port(kind, seconds=2.0, n=4096, chunk=64):
    lo, hi = BOX[kind]
    state = np.random.RandomState(0).uniform(-1, 1, (n, O.STATE_DIM[kind])).astype(np.float32)
    sbd, epi = np.full(n, -1, np.int32), np.ones(n, np.uint32)
    steps, t, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        k, _, _ = O.rollout_random(kind, state, sbd, epi, chunk, 0, 0, lo, hi, t0=t)
        steps += k
        t += chunk
    return steps / (time.perf_counter() - t0)


def main():
    assert R.available(), "needs /root/reference (authoring container only)"
    import ctypes

    O.lib()
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)   # the C oracle's batch loops are OpenMP: ONE thread, like bench.py's cpu_baseline
    out = {"host": f"{os.cpu_count()} logical CPUs, one thread used; measured by oracle/calibrate.py in the authoring container",
           "how_to_read": "ratio = C oracle env-steps/s / reference env-steps/s on the SAME core: divide a cpu_baseline measured elsewhere "
                          "by it to estimate what the reference's own NumPy step() would do there", "kinds": {}}
    for kind in ("quad3d", "quad3d_sl", "quad2d", "quad2d_sl"):
        ref, port = time_reference(kind), time_port(kind)
        out["kinds"][kind] = {"reference_env_steps_per_s": ref, "c_oracle_env_steps_per_s": port, "ratio": port / ref}
        print(kind, f"reference {ref:9.0f}/s   C oracle {port:11.0f}/s   ratio {port / ref:7.1f}", flush=True)
    v, k, el = numpy_ref.time_steps(3.0)
    out["kinds"]["quad3d"]["numpy_restatement_env_steps_per_s"] = v
    out["kinds"]["quad3d"]["numpy_restatement_over_reference"] = v / out["kinds"]["quad3d"]["reference_env_steps_per_s"]
    env = R.RefReinmav()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 4.0:
        env.step()
        n += 1
    ref = n / (time.perf_counter() - t0)
    S, T = np.tile(np.asarray(env.env.state, np.float64).ravel()[None], (64, 1)), np.zeros(64)
    m, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        S, T, _ = O.reinmav_batch_step(S, T)
        m += 64
    port = m / (time.perf_counter() - t0)
    out["kinds"]["reinmav"] = {"reference_env_steps_per_s": ref, "c_oracle_env_steps_per_s": port, "ratio": port / ref}
    print("reinmav", f"reference {ref:9.0f}/s   C oracle {port:11.0f}/s   ratio {port / ref:7.1f}", flush=True)
    path = os.path.join(os.path.dirname(HERE), "profiles", "cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
