"""Per-env NumPy restatement of Quadrotor3D.step / control in the reference's own style (one env, fp64,
small ndarray temporaries, a quaternion helper doing what pyquaternion does).  TEST INFRASTRUCTURE ONLY.

Two uses (SURVEY.md section 8d "CPU baseline timing" (ii)):
* an interpreter-bound stand-in for the reference on the GPU box, where the reference's Python cannot travel:
  bench.py times it for a couple of seconds next to the C oracle;
* a second, independently written oracle: tests check it against the golden vectors and the C oracle.
Line citations refer to gym_reinmav/envs/native/quadrotor3d.py of the reference.
"""
from __future__ import annotations

import numpy as np
from numpy import linalg


def _normalised(q):                       # pyquaternion Quaternion._normalise()
    n2 = float(np.dot(q, q))
    if not abs(1.0 - n2) < 1e-14:
        n = np.sqrt(n2)
        if n > 0:
            return q / n
    return q


def _q_matrix(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])


def _q_bar_matrix(q):
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])


def _rotation_matrix(qn):                 # Quaternion.rotation_matrix (after normalisation)
    return np.dot(_q_matrix(qn), _q_bar_matrix(qn).conj().transpose())[1:][:, 1:]


class Quadrotor3DNumpy:
    def __init__(self):
        self.mass, self.dt = 1.0, 0.01                               # :45-46
        self.g = np.array([0.0, 0.0, -9.8])                          # :47
        self.ref_pos, self.ref_vel = np.array([0.0, 0.0, 2.0]), np.zeros(3)   # :51-52
        self.pos_threshold, self.vel_threshold = 3.0, 10.0           # :55-56
        self.steps_beyond_done = None
        self.state = np.zeros(10)

    def step(self, action):                                          # :81-124
        thrust, w = action[0], action[1:4]
        s = self.state
        pos, att, vel = np.array(s[0:3]), np.array(s[3:7]), np.array(s[7:10])
        qn = _normalised(att)
        acc = thrust / self.mass * _rotation_matrix(qn).dot(np.array([0.0, 0.0, 1.0])) + self.g
        pos = pos + vel * self.dt + 0.5 * acc * self.dt * self.dt
        vel = vel + acc * self.dt
        q_dot = np.dot(_q_matrix(np.dot(_q_matrix(np.array([0.5, 0.0, 0.0, 0.0])), qn)), np.array([0.0, w[0], w[1], w[2]]))
        att = att + q_dot * self.dt
        self.state = np.concatenate([pos, att, vel])
        done = bool(linalg.norm(pos, 2) > self.pos_threshold or linalg.norm(vel, 2) > self.vel_threshold)
        if not done:
            reward = -linalg.norm(pos, 2)
        elif self.steps_beyond_done is None:
            self.steps_beyond_done = 0
            reward = 1.0
        else:
            self.steps_beyond_done += 1
            reward = 0.0
        return np.array(self.state), float(reward), done, {}

    def control(self):                                               # :126-180 (omega is invariant to q_des's sign branch)
        s = self.state
        pos, att, vel = np.array(s[0:3]), np.array(s[3:7]), np.array(s[7:10])
        a_d = -5.0 * (pos - self.ref_pos) + -4.0 * (vel - self.ref_vel) - self.g
        zb = a_d / linalg.norm(a_d)
        xb = np.cross(np.array([0.0, 1.0, 0.0]), zb)
        xb = xb / linalg.norm(xb)
        yb = np.cross(zb, xb)
        R = np.array([xb, yb, zb]).T
        m = R.T                                                      # trace method on the transpose
        if m[2, 2] < 0:
            if m[0, 0] > m[1, 1]:
                t = 1 + m[0, 0] - m[1, 1] - m[2, 2]; q = [m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]]
            else:
                t = 1 - m[0, 0] + m[1, 1] - m[2, 2]; q = [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]]
        else:
            if m[0, 0] < -m[1, 1]:
                t = 1 - m[0, 0] - m[1, 1] + m[2, 2]; q = [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t]
            else:
                t = 1 + m[0, 0] + m[1, 1] + m[2, 2]; q = [t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]]
        q_des = np.array(q) * (0.5 / np.sqrt(t))
        conj = np.array([att[0], -att[1], -att[2], -att[3]])
        qe = np.dot(_q_matrix(conj), q_des)
        w = (2 / 0.3) * np.sign(qe[0]) * qe[1:4]
        thrust = a_d.dot(_rotation_matrix(_normalised(att)).dot(np.array([0.0, 0.0, 1.0])))
        return np.array([thrust, w[0], w[1], w[2]])


def time_steps(seconds: float = 2.0, seed: int = 0):
    """env-steps/s of the per-env NumPy loop: random actions, reset on done (what a python-level env costs)."""
    import time

    rng = np.random.RandomState(seed)
    env = Quadrotor3DNumpy()
    env.state = rng.uniform(-1, 1, 10)
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(200):
            _, _, done, _ = env.step(rng.uniform(0, 10, 4))
            if done:
                env.state = rng.uniform(-1, 1, 10)
        n += 200
        el = time.perf_counter() - t0
        if el >= seconds:
            return n / el, n, el
