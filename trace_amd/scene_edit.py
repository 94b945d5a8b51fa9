"""The caller of the sampler: guided closed-loop rollout and heuristic guidance configuration.

Mirrors /root/reference/tbsim/utils/scene_edit_utils.py (same function names, arguments and
returned structures):
  * ``guided_rollout``             -- :9-111
  * ``merge_guidance_configs``     -- :117-126
  * the heuristics                 -- :181-357, ``HEURISTIC_FUNC`` :359-367
  * ``compute_heuristic_guidance`` -- :369-392

The environment is any object with the reference's gym-like surface; here that is
``trace_amd.sim_env.SceneBatchEnv`` (device-resident), whose observations are already torch tensors
on the policy's device, so ``obs_to_torch`` only moves what is not there yet.  The per-guidance
success metrics of the reference (``guidance_metrics_from_config``, shapely-based) are evaluation
reporting and not built: ``guidance_metrics_fn`` lets a caller supply them.
"""
import random
from copy import deepcopy

import numpy as np
import torch


def _to_torch(x, device):
    if isinstance(x, dict):
        return {k: _to_torch(v, device) for k, v in x.items()}
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).to(device)
    if torch.is_tensor(x):
        return x.to(device)
    return x


def guided_rollout(env, policy, policy_model, n_step_action=1, guidance_config=None, scene_indices=None,
                   device=None, obs_to_torch=True, horizon=None, use_gt=False, start_frames=None,
                   guidance_metrics_fn=None, policy_kwargs=None):
    """Roll the environment out under ``policy`` (a ``RolloutWrapper``), replanning every
    ``n_step_action`` frames.  Returns (stats, info) exactly as the reference: per-metric arrays
    concatenated over the scenes of the batch, and the environment's info lists."""
    stats, info = {}, {}
    added_metrics = []
    policy_kwargs = policy_kwargs or {}
    if guidance_config is not None:
        # an example batch initialises the guidance (scene_edit_utils.py:47-58)
        env.reset(scene_indices=scene_indices, start_frame_index=start_frames)
        ex_obs = env.get_observation()
        if obs_to_torch:
            device = policy.device if device is None else device
            ex_obs = _to_torch(ex_obs, device)
        if not use_gt:
            policy_model.set_guidance(guidance_config, ex_obs["agents"])
        if guidance_metrics_fn is None:
            from .guidance_metrics import guidance_metrics_from_config as guidance_metrics_fn
        if guidance_metrics_fn:
            gm = guidance_metrics_fn(guidance_config)
            env._metrics.update(gm)
            added_metrics += list(gm.keys())

    env.reset(scene_indices=scene_indices, start_frame_index=start_frames)
    done = env.is_done()
    counter = 0
    while not done:
        obs = env.get_observation()
        if obs_to_torch:
            device = policy.device if device is None else device
            obs = _to_torch(obs, device)
        action = policy.get_action(obs, step_index=counter, **policy_kwargs)
        env.step(action, num_steps_to_take=n_step_action, render=False)
        counter += n_step_action
        done = env.is_done()
        if horizon is not None and counter >= horizon:
            break

    for k, v in env.get_metrics().items():
        stats[k] = np.concatenate([stats[k], v], axis=0) if k in stats else np.asarray(v)
    for name in added_metrics:
        env._metrics.pop(name)
    if not use_gt:
        policy_model.clear_guidance()
    for k, v in env.get_info().items():
        info.setdefault(k, []).extend(v)
    env.reset_multi_episodes_metrics()
    return stats, info


def merge_guidance_configs(cfg1, cfg2):
    if cfg1 is None or len(cfg1) == 0:
        return cfg2
    if cfg2 is None or len(cfg2) == 0:
        return cfg1
    merged = deepcopy(cfg1)
    for si in range(len(merged)):
        merged[si].extend(cfg2[si])
    return merged


# ---------------------------------------------------------------------------------------------- #
# heuristics: scene state -> guidance dicts {name, params, agents} (the weight is added by the caller)
# ---------------------------------------------------------------------------------------------- #
def get_agents_future(sim_scene, fut_sec):
    return sim_scene.get_agents_future(fut_sec)


def get_agents_curr(sim_scene):
    return sim_scene.get_agents_curr()


def get_agent_from_world_tf(curr_state):
    pos, h = curr_state[:, :2], curr_state[:, -1]
    c, s = np.cos(h), np.sin(h)
    wfa = np.zeros((pos.shape[0], 3, 3))
    wfa[:, 0, 0], wfa[:, 0, 1], wfa[:, 0, 2] = c, -s, pos[:, 0]
    wfa[:, 1, 0], wfa[:, 1, 1], wfa[:, 1, 2] = s, c, pos[:, 1]
    wfa[:, 2, 2] = 1.0
    return np.linalg.inv(wfa)


def _last_valid_targets(sim_scene, dt, target_time, perturb_std):
    """Target = recorded position at the latest valid frame up to ``target_time`` (:235-256)."""
    fut, valid = get_agents_future(sim_scene, target_time * dt)
    fut_pos = fut[:, :, :2]
    agents = np.arange(fut_pos.shape[0])
    seen = valid.sum(-1) > 0
    if seen.sum() == 0:
        return None
    fut_pos, valid, agents = fut_pos[seen], valid[seen], agents[seen]
    n, t = valid.shape
    last_t = np.amax(np.arange(t)[None].repeat(n, 0) * valid, axis=-1)
    target = fut_pos[np.arange(n), last_t]
    if perturb_std is not None and perturb_std > 0.0:
        target = target + np.random.randn(*target.shape) * perturb_std
    return target, last_t, agents, seen


def heuristic_global_target_pos_at_time(sim_scene, dt, target_time, urgency, pref_speed, perturb_std=None):
    r = _last_valid_targets(sim_scene, dt, target_time, perturb_std)
    if r is None:
        return None
    target, last_t, agents, _ = r
    return {"name": "global_target_pos_at_time",
            "params": {"target_pos": target.tolist(), "target_time": last_t.tolist(),
                       "urgency": [urgency] * len(agents), "pref_speed": pref_speed, "dt": dt},
            "agents": agents.tolist()}


def heuristic_global_target_pos(sim_scene, dt, target_time, urgency, pref_speed, min_progress_dist,
                                perturb_std=None):
    g = heuristic_global_target_pos_at_time(sim_scene, dt, target_time, urgency, pref_speed, perturb_std)
    g["name"] = "global_target_pos"
    g["params"]["min_progress_dist"] = min_progress_dist
    g["params"].pop("target_time", None)
    return g


def heuristic_target_pos_at_time(sim_scene, dt, target_time, perturb_std=None):
    r = _last_valid_targets(sim_scene, dt, target_time, perturb_std)
    if r is None:
        return None
    target, last_t, agents, seen = r
    afw = get_agent_from_world_tf(get_agents_curr(sim_scene)[seen])
    local = np.einsum("nij,nj->ni", afw[:, :2, :2], target) + afw[:, :2, 2]
    return {"name": "target_pos_at_time",
            "params": {"target_pos": local.tolist(), "target_time": last_t.tolist()},
            "agents": agents.tolist()}


def heuristic_target_pos(sim_scene, dt, target_time, perturb_std=None):
    g = heuristic_target_pos_at_time(sim_scene, dt, target_time, perturb_std)
    g["name"] = "target_pos"
    g["params"].pop("target_time", None)
    return g


def heuristic_agent_collision(sim_scene, dt, num_disks, buffer_dist):
    return {"name": "agent_collision", "params": {"num_disks": num_disks, "buffer_dist": buffer_dist},
            "agents": None}


def heuristic_map_collision(sim_scene, dt, num_points_lw):
    return {"name": "map_collision", "params": {"num_points_lw": num_points_lw}, "agents": None}


def heuristic_social_group(sim_scene, dt, group_dist_thresh, social_dist, cohesion):
    """Groups = connected components of the graph "closer than ``group_dist_thresh`` and not walking in
    opposing directions" (:181-229); one ``social_group`` guide per group of two or more, with a
    random leader."""
    from scipy.sparse.csgraph import connected_components
    cur = get_agents_curr(sim_scene)
    pos, vel = cur[:, :2], cur[:, 2:4]
    speed = np.linalg.norm(vel, axis=-1)
    still = speed < 0.9
    d = vel / (speed[:, None] + 1e-6)
    same_way = (d @ d.T) >= 0
    same_way[still] = True
    same_way[:, still] = True
    dist = np.linalg.norm(pos[:, None] - pos[None], axis=-1)
    graph = np.logical_and(dist <= group_dist_thresh, same_way)
    np.fill_diagonal(graph, False)
    n_comp, labels = connected_components(graph.astype(int), directed=False)
    out = []
    for ci in range(n_comp):
        members = np.nonzero(labels == ci)[0].tolist()
        if len(members) > 1:
            out.append({"name": "social_group",
                        "params": {"leader_idx": random.sample(members, 1)[0], "social_dist": social_dist,
                                   "cohesion": cohesion},
                        "agents": members})
    return out if out else None


HEURISTIC_FUNC = {
    "global_target_pos_at_time": heuristic_global_target_pos_at_time,
    "global_target_pos": heuristic_global_target_pos,
    "target_pos_at_time": heuristic_target_pos_at_time,
    "target_pos": heuristic_target_pos,
    "agent_collision": heuristic_agent_collision,
    "map_collision": heuristic_map_collision,
    "social_group": heuristic_social_group,
}


def compute_heuristic_guidance(heuristic_config, env, scene_indices, start_frames):
    """One guidance list per scene from the heuristic entries ``{name, weight, params}``.  Unlike
    the reference (:382, which pops ``dt`` out of the caller's params dict on the first scene), the
    caller's config is left untouched."""
    env.reset(scene_indices=scene_indices, start_frame_index=start_frames)
    out = []
    for i, _ in enumerate(scene_indices):
        scene = env._current_scenes[i]
        dt = scene.dataset.desired_dt
        guides = []
        for h in heuristic_config:
            assert {"name", "weight", "params"}.issubset(h.keys()), "All heuristics must have these 3 fields"
            assert h["name"] in HEURISTIC_FUNC, "Unrecognized heuristic!"
            params = dict(h["params"])
            dt = params.pop("dt", dt)
            g = HEURISTIC_FUNC[h["name"]](scene, dt, **params)
            if g is not None:
                for el in (g if isinstance(g, list) else [g]):
                    el["weight"] = h["weight"]
                    guides.append(el)
        out.append(guides)
    return out
