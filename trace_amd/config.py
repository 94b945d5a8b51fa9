"""Model hyper-parameters of the TRACE sampler path.

The defaults are the "algo" block of the reference's orca_mixed training config
(/root/reference/configs/train/diffuser_orca_mlp_hist_grid_map_drop_10.json:90-206),
which is what ``configs/eval/orca/*.json`` checkpoints are trained with.  A config
can also be built from such a JSON (``ModelConfig.from_algo_dict``).
"""
import dataclasses
import json
import math
from typing import List, Tuple


@dataclasses.dataclass
class ModelConfig:
    horizon: int = 52
    n_diffusion_steps: int = 100
    base_dim: int = 32
    dim_mults: Tuple[int, ...] = (2, 4, 8)
    cond_feat_dim: int = 256
    map_grid_feature_dim: int = 32
    history_feature_dim: int = 128
    history_num_frames: int = 30
    step_time: float = 0.1
    conditioning_drop_fill: float = 0.5
    # Unicycle bounds (reference tbsim/dynamics/unicycle.py:7-18 defaults vbound;
    # acce_bound / max_steer / max_yawvel from the algo config "dynamics" block)
    max_steer: float = 0.5
    max_yawvel: float = 2.0 * math.pi
    acce_bound: Tuple[float, float] = (-10.0, 8.0)
    vbound: Tuple[float, float] = (-10.0, 30.0)
    # x, y, vel, yaw, acc, yawvel :  scaled = (x + add) / div
    norm_add: Tuple[float, ...] = (-3.538049, 0.004175, -1.360894, 0.001894, 0.015233, 0.000562)
    norm_div: Tuple[float, ...] = (2.304491, 0.462847, 0.426683, 0.19193, 0.255089, 0.175583)
    agent_hist_norm_add: Tuple[float, ...] = (2.07473, -0.000102, -1.401319, 0.0, 0.0)
    agent_hist_norm_div: Tuple[float, ...] = (1.376582, 0.214366, 0.369064, 1.0, 1.0)
    neighbor_hist_norm_add: Tuple[float, ...] = (-2.804952, -0.037075, -0.544995, 0.0, 0.0)
    neighbor_hist_norm_div: Tuple[float, ...] = (6.161589, 6.486891, 0.771583, 1.0, 1.0)
    # raster geometry (input image C,H,W; feature grid is H/4 x W/4)
    image_shape: Tuple[int, int, int] = (2, 224, 224)
    grid_hw: Tuple[int, int] = (56, 56)
    n_groups: int = 8

    @property
    def transition_dim(self):
        return 6

    @property
    def unet_in_dim(self):
        return self.transition_dim + self.map_grid_feature_dim

    @property
    def unet_dims(self) -> List[int]:
        return [self.unet_in_dim] + [self.base_dim * m for m in self.dim_mults]

    @property
    def time_dim(self):
        return self.base_dim

    @property
    def embed_dim(self):
        return self.cond_feat_dim + self.time_dim

    @property
    def hist_frames(self):
        return self.history_num_frames + 1

    @staticmethod
    def from_algo_dict(algo: dict, image_shape=(2, 224, 224)) -> "ModelConfig":
        dyn = algo["dynamics"]
        return ModelConfig(
            horizon=int(algo["horizon"]), n_diffusion_steps=int(algo["n_diffusion_steps"]),
            base_dim=int(algo["base_dim"]), dim_mults=tuple(algo["dim_mults"]),
            cond_feat_dim=int(algo["cond_feat_dim"]),
            map_grid_feature_dim=int(algo["map_grid_feature_dim"]),
            history_feature_dim=int(algo["history_feature_dim"]),
            history_num_frames=int(algo["history_num_frames"]),
            step_time=float(algo["step_time"]),
            conditioning_drop_fill=float(algo["conditioning_drop_fill"]),
            max_steer=float(dyn["max_steer"]), max_yawvel=float(dyn["max_yawvel"]),
            acce_bound=tuple(float(v) for v in dyn["acce_bound"]),
            norm_add=tuple(algo["diffuser_norm_info"][0]), norm_div=tuple(algo["diffuser_norm_info"][1]),
            agent_hist_norm_add=tuple(algo["agent_hist_norm_info"][0]),
            agent_hist_norm_div=tuple(algo["agent_hist_norm_info"][1]),
            neighbor_hist_norm_add=tuple(algo["neighbor_hist_norm_info"][0]),
            neighbor_hist_norm_div=tuple(algo["neighbor_hist_norm_info"][1]),
            image_shape=tuple(image_shape),
            grid_hw=(image_shape[1] // 4, image_shape[2] // 4),
        )

    @staticmethod
    def from_json(path: str, image_shape=(2, 224, 224)) -> "ModelConfig":
        with open(path) as f:
            cfg = json.load(f)
        return ModelConfig.from_algo_dict(cfg["algo"] if "algo" in cfg else cfg, image_shape)

    def to_algo_dict(self) -> dict:
        """The inverse of from_algo_dict (used by the golden-vector generator to
        instantiate the reference with the same hyper-parameters)."""
        return {
            "horizon": self.horizon, "n_diffusion_steps": self.n_diffusion_steps,
            "base_dim": self.base_dim, "dim_mults": list(self.dim_mults),
            "cond_feat_dim": self.cond_feat_dim, "map_feature_dim": 256,
            "map_grid_feature_dim": self.map_grid_feature_dim,
            "history_feature_dim": self.history_feature_dim,
            "history_num_frames": self.history_num_frames, "step_time": self.step_time,
            "conditioning_drop_fill": self.conditioning_drop_fill,
            "dynamics": {"type": "Unicycle", "max_steer": self.max_steer,
                         "max_yawvel": self.max_yawvel, "acce_bound": list(self.acce_bound)},
            "diffuser_norm_info": [list(self.norm_add), list(self.norm_div)],
            "agent_hist_norm_info": [list(self.agent_hist_norm_add), list(self.agent_hist_norm_div)],
            "neighbor_hist_norm_info": [list(self.neighbor_hist_norm_add),
                                        list(self.neighbor_hist_norm_div)],
        }


def unet_block_table(cfg: ModelConfig):
    """Residual blocks of the temporal UNet in execution order
    (reference tbsim/models/temporal.py:78-115,117-142).
    Returns list of (state_dict prefix, c_in, c_out, length)."""
    dims = cfg.unet_dims
    in_out = list(zip(dims[:-1], dims[1:]))
    n_res = len(in_out)
    L = cfg.horizon
    blocks = []
    for i, (ci, co) in enumerate(in_out):
        blocks.append(("model.downs.%d.0" % i, ci, co, L))
        blocks.append(("model.downs.%d.1" % i, co, co, L))
        if i < n_res - 1:
            L = L // 2
    mid = dims[-1]
    blocks.append(("model.mid_block1", mid, mid, L))
    blocks.append(("model.mid_block2", mid, mid, L))
    for i, (ci, co) in enumerate(reversed(in_out[1:])):
        blocks.append(("model.ups.%d.0" % i, co * 2, ci, L))
        blocks.append(("model.ups.%d.1" % i, ci, ci, L))
        L = L * 2
    return blocks
