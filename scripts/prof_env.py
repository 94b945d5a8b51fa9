"""Environment-side kernels at the cfg5 per-GPU shape (4096 agents), for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.config import ModelConfig          # noqa: E402
from trace_amd.sim_env import SceneBatchEnv       # noqa: E402

cfg = ModelConfig()
env = SceneBatchEnv(cfg, [32] * 128, device="cuda:0", seed=1)
env.reset()
pose = env._hist[:, -1]
for _ in range(5):
    img, drv = env._rasterize(pose)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    env._cached_observation = None
    obs = env.get_observation()
e1.record()
torch.cuda.synchronize()
C, H, W = cfg.image_shape
print("get_observation: %.2f ms for %d agents (raster bytes written per call: %.0f MB)"
      % (e0.elapsed_time(e1) / 10, pose.shape[0], pose.shape[0] * H * W * (4 * C + 1) / 1e6))
