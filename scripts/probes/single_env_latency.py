"""Steps per second of the reference-shaped single env (gym API, N = 1): host round trips dominate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gym_sbr2_amd as g
env = g.make("SBROS-v1")
for ep in range(3):
    env.reset()
    t0 = time.perf_counter(); n = 0; done = False
    while not done:
        obs, state, r, done, info = env.step([2.0, 5.0]); n += 1
    dt = time.perf_counter() - t0
    print("episode %d: %d calls in %.1f ms = %.1f us per step = %.0f steps/s" % (ep, n, dt * 1e3, dt / n * 1e6, n / dt))
