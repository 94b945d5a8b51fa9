import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from trace_amd.config import ModelConfig
from trace_amd.sim_env import SceneBatchEnv
cfg = ModelConfig()
env = SceneBatchEnv(cfg, [32] * 128, device="cuda:0", seed=1)
env.reset()
pose = env._hist[:, -1]
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("rasterize %.2f ms" % t(lambda: env._rasterize(pose)))
env.with_image = False
def obs():
    env._cached_observation = None
    return env.get_observation()
print("obs without image %.2f ms" % t(obs))
env.with_image = True
print("obs with image %.2f ms" % t(obs))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); obs(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
