"""Policy boundary of the sampler: the objects the reference's evaluation driver talks to.

Mirrors (same names, argument meaning and return structure):
  * ``Action`` / ``RolloutAction``      -- /root/reference/tbsim/policies/common.py:8-64,107-180
  * ``PolicyWrapper`` / ``RolloutWrapper`` -- /root/reference/tbsim/policies/wrappers.py:6-65
  * ``DiffuserTrafficModel.get_action / set_guidance / clear_guidance``
                                          -- /root/reference/tbsim/algos/algos.py:261-326
  * ``Diffuser`` composer               -- /root/reference/tbsim/evaluation/policy_composers.py:27-57

``trace_amd/compat/tbsim/`` re-exports these under the reference's import paths
(``tbsim.evaluation.policy_composers.Diffuser`` ...) when ``trace_amd/compat`` is put on
PYTHONPATH, which is how a driver written like ``scripts/scene_editor.py --eval_class Diffuser``
resolves them (scene_editor.py:48-51).
"""
import json
import os

import numpy as np
import torch

from .config import ModelConfig
from .map_encoder import MapEncoder, has_map_encoder
from .sampler import TraceSampler
from .weights import extract_policy_state_dict, random_state_dict


def _clone(x):
    return x.clone() if torch.is_tensor(x) else np.array(x, copy=True)


def _to_numpy(x):
    if isinstance(x, dict):
        return {k: _to_numpy(v) for k, v in x.items()}
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return x


class Action(object):
    """Positions (…,T,2) and yaws (…,T,1) of the chosen plan."""

    def __init__(self, positions, yaws):
        assert positions.shape[:-1] == yaws.shape[:-1]
        assert positions.shape[-1] == 2 and yaws.shape[-1] == 1
        self._positions, self._yaws = positions, yaws

    @property
    def positions(self):
        return _clone(self._positions)

    @property
    def yaws(self):
        return _clone(self._yaws)

    @property
    def trajectories(self):
        if isinstance(self._positions, np.ndarray):
            return np.concatenate([self._positions, self._yaws], axis=-1)
        return torch.cat([self._positions, self._yaws], dim=-1)

    def to_dict(self):
        return dict(positions=self.positions, yaws=self.yaws)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    def to_numpy(self):
        return self.__class__(**_to_numpy(self.to_dict()))


class RolloutAction(object):
    def __init__(self, ego=None, ego_info=None, agents=None, agents_info=None):
        assert ego is None or isinstance(ego, Action)
        assert agents is None or isinstance(agents, Action)
        self.ego, self.ego_info, self.agents, self.agents_info = ego, ego_info, agents, agents_info

    @property
    def has_ego(self):
        return self.ego is not None

    @property
    def has_agents(self):
        return self.agents is not None

    def to_numpy(self):
        return self.__class__(
            ego=self.ego.to_numpy() if self.has_ego else None,
            ego_info=_to_numpy(self.ego_info) if self.has_ego else None,
            agents=self.agents.to_numpy() if self.has_agents else None,
            agents_info=_to_numpy(self.agents_info) if self.has_agents else None)


class PolicyWrapper(object):
    """Binds run-time keyword arguments of ``get_action`` (wrappers.py:6-30)."""

    def __init__(self, model, get_action_kwargs=None, get_plan_kwargs=None):
        self.model = model
        self.device = model.device
        self.action_kwargs = get_action_kwargs or {}
        self.plan_kwargs = get_plan_kwargs

    def eval(self):
        self.model.eval()

    def get_action(self, obs, **kwargs):
        return self.model.get_action(obs, **self.action_kwargs, **kwargs)

    @classmethod
    def wrap_controller(cls, model, **kwargs):
        return cls(model=model, get_action_kwargs=kwargs)


class RolloutWrapper(object):
    """wrappers.py:33-65 (agents policy only is the path TRACE uses)."""

    def __init__(self, ego_policy=None, agents_policy=None, pass_agent_obs=True):
        self.device = ego_policy.device if agents_policy is None else agents_policy.device
        self.ego_policy, self.agents_policy, self.pass_agent_obs = ego_policy, agents_policy, pass_agent_obs

    def eval(self):
        for p in (self.ego_policy, self.agents_policy):
            if p is not None:
                p.eval()

    def get_action(self, obs, step_index, **agents_kwargs):
        """``agents_kwargs`` (not in the reference) reach the agents policy: noise injection."""
        ego = ego_info = agents = agents_info = None
        with torch.no_grad():
            if self.ego_policy is not None:
                assert obs["ego"] is not None
                kw = dict(agent_obs=obs["agents"]) if self.pass_agent_obs else {}
                ego, ego_info = self.ego_policy.get_action(obs["ego"], step_index=step_index, **kw)
            if self.agents_policy is not None:
                assert obs["agents"] is not None
                agents, agents_info = self.agents_policy.get_action(obs["agents"], step_index=step_index,
                                                                    **agents_kwargs)
        return RolloutAction(ego, ego_info, agents, agents_info)


def choose_action_from_gt(preds, obs_dict):
    """Sample closest to the ground-truth future (guidance_loss.py:43-65)."""
    B, N, T, _ = preds["positions"].shape
    idx = torch.zeros(B, dtype=torch.long, device=preds["positions"].device)
    if "target_positions" not in obs_dict:
        return idx
    end = min(T, obs_dict["target_positions"].shape[1])
    gt = obs_dict["target_positions"][:, :end].unsqueeze(1).to(preds["positions"].device)
    valid = obs_dict["target_availabilities"][:, :end].unsqueeze(1).expand(B, N, end).to(gt.device)
    err = torch.norm(preds["positions"][:, :, :end] - gt, dim=-1)
    err = torch.where(valid.bool(), err, torch.full_like(err, float("nan")))
    ade = torch.nanmean(err, dim=-1)
    ok = torch.isnan(ade).sum(dim=-1) == 0
    best = torch.argmin(torch.nan_to_num(ade, nan=float("inf")), dim=-1)
    return torch.where(ok, best, idx)


class GTNaNPolicy(object):
    """Replays the recorded future of the observation as the plan; unavailable frames are NaN
    (/root/reference/tbsim/policies/hardcoded.py:8-29; ``scripts/scene_editor.py --ground_truth``)."""

    def __init__(self, device):
        self.device = torch.device(device)

    def eval(self):
        pass

    def get_action(self, obs, **kwargs):
        invalid = ~torch.as_tensor(obs["target_availabilities"]).bool().to(self.device)
        pos = torch.as_tensor(obs["target_positions"]).to(self.device, torch.float32).clone()
        yaw = torch.as_tensor(obs["target_yaws"]).to(self.device, torch.float32).clone()
        pos[invalid] = float("nan")
        yaw[invalid] = float("nan")
        return Action(positions=pos, yaws=yaw), {}


class DiffuserTrafficModel(object):
    """Algo-level object with the reference's inference interface (algos.py:19-326).  Holds one
    ``TraceSampler`` (the EMA copy is what inference uses, algos.py:268-271)."""

    def __init__(self, model_config: ModelConfig, state_dict, device="cuda:0", guidance_config=None,
                 map_encoder=None):
        self.algo_config = model_config
        self.device = torch.device(device)
        if map_encoder is None and has_map_encoder(state_dict):
            # the checkpoint carries the ResNet-18 map encoder (trace.py:103-113): image -> grid
            map_encoder = MapEncoder(state_dict, device)
        self.sampler = TraceSampler(model_config, state_dict, device, map_encoder=map_encoder)
        if guidance_config is not None:
            self.set_guidance(guidance_config)

    @classmethod
    def load_from_checkpoint(cls, ckpt_path, model_config, device="cuda:0", use_ema=True, **kw):
        """Loads a Lightning ``.ckpt`` written by the reference's training.
        Only tensors are read (``weights_only=True``)."""
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        return cls(model_config, extract_policy_state_dict(sd, use_ema), device, **kw)

    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device) == self.device, "construct the model on its target device"
        return self

    def set_guidance(self, guidance_config, example_batch=None):
        self.sampler.set_guidance(guidance_config, example_batch)

    def clear_guidance(self):
        self.sampler.clear_guidance()

    def forward(self, obs_dict, num_samp=1, class_free_guide_w=0.0, guide_as_filter_only=False,
                guide_clean=False, **sampler_kwargs):
        return self.sampler(obs_dict, num_samp, return_diffusion=True, return_guidance_losses=True,
                            class_free_guide_w=class_free_guide_w,
                            apply_guidance=(not guide_as_filter_only), guide_clean=guide_clean,
                            **sampler_kwargs)

    __call__ = forward

    def get_action(self, obs_dict, num_action_samples=1, class_free_guide_w=0.0,
                   guide_as_filter_only=False, guide_with_gt=False, guide_clean=False, **kwargs):
        step_index = kwargs.pop("step_index")
        self.sampler.update_guidance(global_t=step_index)
        sampler_kwargs = {k: kwargs[k] for k in ("noise_init", "noise_steps", "generator", "use_graph")
                          if k in kwargs}
        out = self.forward(obs_dict, num_samp=num_action_samples, class_free_guide_w=class_free_guide_w,
                           guide_as_filter_only=guide_as_filter_only, guide_clean=guide_clean,
                           **sampler_kwargs)
        preds = out["predictions"]
        B = preds["positions"].shape[0]
        act_idx = torch.zeros(B, dtype=torch.long, device=preds["positions"].device)
        if guide_with_gt and "target_positions" in obs_dict:
            act_idx = choose_action_from_gt(preds, obs_dict)
        guide_scores = None
        if self.sampler.current_guidance is not None:
            guide_scores = out["guidance_plan"].total_scores(out["guidance_layer_losses"])
        if guide_with_gt and "target_positions" in obs_dict:
            pass
        elif self.sampler.current_guidance is not None:
            act_idx = out["guidance_plan"].choose_action(out["guidance_layer_losses"])
        ar = torch.arange(B, device=act_idx.device)
        info = dict(
            action_samples=Action(positions=preds["positions"], yaws=preds["yaws"]).to_dict(),
            diffusion_steps={"traj": preds["diffusion_steps"][ar, act_idx]},
            act_idx=act_idx,
        )
        if "guide_losses" in preds:
            info["guide_losses"] = preds["guide_losses"]
        if guide_scores is not None:
            info["guide_scores"] = guide_scores
        action = Action(positions=preds["positions"][ar, act_idx], yaws=preds["yaws"][ar, act_idx])
        return action, info


def get_checkpoint(ckpt_key, ckpt_dir):
    """Checkpoint and config paths inside ``ckpt_dir`` (experiment_utils.py:39-80): the one ``*.ckpt``
    anywhere below it whose path contains ``ckpt_key`` (``FileNotFoundError`` for none,
    ``AssertionError`` for several) and the first ``config.json`` below it, else the parent's."""
    import glob
    assert ckpt_dir is not None
    ckpts = [p for p in glob.glob(ckpt_dir + "/**/*.ckpt", recursive=True) if str(ckpt_key) in p]
    if len(ckpts) == 0:
        raise FileNotFoundError("Cannot find checkpoint in {} with key {}".format(ckpt_dir, ckpt_key))
    assert len(ckpts) == 1, "More than one checkpoint found {}".format(ckpts)
    cfgs = sorted(glob.glob(ckpt_dir + "/**/config.json", recursive=True)) or glob.glob(ckpt_dir + "/../config.json")
    if len(cfgs) == 0:
        raise FileNotFoundError("no config.json in or next to {}".format(ckpt_dir))
    return ckpts[0], cfgs[0]


def modality_image_shape(exp_cfg):
    """(C, H, W) of the rasterised map input from the training config
    (trajdata_utils.py:333-339: history layers when ``include_hist`` + semantic layers)."""
    try:
        r = exp_cfg["env"]["rasterizer"]
        hist = (int(exp_cfg["algo"]["history_num_frames"]) + 1) if r.get("include_hist", False) else 0
        return (hist + int(r["num_sem_layers"]), int(r["raster_size"]), int(r["raster_size"]))
    except (KeyError, TypeError):
        return (2, 224, 224)


class Diffuser(object):
    """Policy composer selected by ``eval_cfg.eval_class`` (policy_composers.py:27-57).
    ``eval_config`` needs ``.policy.{num_action_samples, class_free_guide_w, guide_as_filter_only,
    guide_with_gt, guide_clean}`` and ``.ckpt.policy.{ckpt_dir, ckpt_key}`` (attribute or dict
    access).  Without a checkpoint on disk (none ship with the reference) it falls back to
    seeded random weights of the orca_mixed architecture when ``allow_random_init`` is set."""

    def __init__(self, eval_config, device, ckpt_root_dir=None, allow_random_init=False,
                 seed=0, random_map_encoder=False):
        if ckpt_root_dir is None:
            try:
                ckpt_root_dir = self._get(eval_config, "ckpt_root_dir")
            except (KeyError, AttributeError):
                ckpt_root_dir = "checkpoints/"
        self.eval_config, self.device, self.ckpt_root_dir = eval_config, device, ckpt_root_dir
        self.allow_random_init, self.seed = allow_random_init, seed
        self.random_map_encoder = random_map_encoder

    @staticmethod
    def _get(cfg, *path):
        for p in path:
            cfg = cfg[p] if isinstance(cfg, dict) else getattr(cfg, p)
        return cfg

    def _find_checkpoint(self):
        try:
            ckpt_dir = self._get(self.eval_config, "ckpt", "policy", "ckpt_dir")
            ckpt_key = self._get(self.eval_config, "ckpt", "policy", "ckpt_key")
        except (KeyError, AttributeError):
            return None, None
        if not ckpt_dir:
            return None, None
        # the reference passes ckpt_dir as given (policy_composers.py:36-40); a path relative to
        # ckpt_root_dir is accepted as well
        for root in (ckpt_dir, os.path.join(self.ckpt_root_dir, ckpt_dir)):
            if os.path.isdir(root):
                try:
                    return get_checkpoint(ckpt_key, root)
                except FileNotFoundError:
                    if not self.allow_random_init:
                        raise
        return None, None

    def get_policy(self, policy=None):
        policy_cfg = None
        if policy is None:
            ckpt, cfg_path = self._find_checkpoint()
            if ckpt is not None and cfg_path is not None:
                with open(cfg_path) as f:
                    policy_cfg = json.load(f)
                mc = ModelConfig.from_algo_dict(policy_cfg["algo"], image_shape=modality_image_shape(policy_cfg))
                policy = DiffuserTrafficModel.load_from_checkpoint(ckpt, mc, self.device)
            elif self.allow_random_init:
                mc = ModelConfig()
                sd = random_state_dict(mc, self.seed)
                if self.random_map_encoder:
                    from .map_encoder import random_map_encoder_state_dict
                    sd.update(random_map_encoder_state_dict(mc.image_shape[0], mc.map_grid_feature_dim,
                                                            mc.image_shape[1:], seed=self.seed))
                policy = DiffuserTrafficModel(mc, sd, self.device)
                # the parts of the experiment config the driver reads (scene_editor.py:61,107,227)
                policy_cfg = {"algo": mc.to_algo_dict(),
                              "env": {"rasterizer": {"include_hist": False, "num_sem_layers": mc.image_shape[0],
                                                     "drivable_layers": None, "raster_size": mc.image_shape[1],
                                                     "pixel_size": 1.0 / 12.0, "ego_center": [-0.5, 0.0],
                                                     "no_map_fill_value": -1.0}}}
            else:
                raise FileNotFoundError("no TRACE checkpoint found under %s" % self.ckpt_root_dir)
        pc = self._get(self.eval_config, "policy")
        g = (lambda k: pc[k]) if isinstance(pc, dict) else (lambda k: getattr(pc, k))
        wrapped = PolicyWrapper.wrap_controller(
            policy, num_action_samples=g("num_action_samples"),
            class_free_guide_w=g("class_free_guide_w"),
            guide_as_filter_only=g("guide_as_filter_only"), guide_with_gt=g("guide_with_gt"),
            guide_clean=g("guide_clean"))
        return wrapped, policy_cfg
