import sys, time
sys.path.insert(0, "reinmav-gym_amd")
import gym_reinmav_amd as g
for flag in (1, 0, 1, 0):
    env = g.make("quadrotor2d-v0")
    env._batch.set_tuning(host_flag=flag)
    env.reset()
    for _ in range(400):
        a = env.control(); _, _, d, _ = env.step(a)
        if d: env.reset()
    t0 = time.perf_counter(); n = 0
    for ep in range(10):
        env.reset()
        for _ in range(400):
            a = env.control(); _, _, d, _ = env.step(a); n += 1
            if d: env.reset()
    print("host_flag", flag, "us per control+step", 1e6 * (time.perf_counter() - t0) / n)
    env.close()
