"""CPU tests of the callers either side of the sampler: the
device-resident scene environment against the per-agent numpy oracle (oracle/env_oracle.py, which
restates env_trajdata.py:376-397 and the agent-centric convention of trajdata_utils.py:155-273),
the guided rollout loop, the heuristic guidance configuration and the eval-config loader."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import env_oracle as eo
from trace_amd import scene_edit as se
from trace_amd.config import ModelConfig
from trace_amd.eval_config import Dict, SceneEditingConfig, finalize_eval_config
from trace_amd.policy import Action, RolloutAction, RolloutWrapper
from trace_amd.sim_env import CollisionRate, OffRoadRate, SceneBatchEnv

REF_CFG_DIR = "/root/reference/configs/eval"


def small_cfg():
    return ModelConfig(image_shape=(2, 32, 32), grid_hw=(8, 8))


def make_env(**kw):
    args = dict(agents_per_scene=[5, 3, 4], device="cpu", seed=3, num_simulation_steps=20, max_neighbors=3,
                pixel_size=1.0 / 2.0)
    args.update(kw)
    return SceneBatchEnv(small_cfg(), **args)


def random_action(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.cumsum(torch.rand(B, T, 2, generator=g) * 0.2, 1)
    yaw = (torch.rand(B, T, 1, generator=g) - 0.5) * 0.6
    return RolloutAction(agents=Action(pos, yaw), agents_info={})


def test_step_matches_reference_formula():
    env = make_env()
    env.reset()
    obs = env.get_observation()["agents"]
    B = obs["centroid"].shape[0]
    act = random_action(B, 12, 0)
    cur = obs["curr_agent_state"].numpy()
    want = eo.apply_action(cur, act.agents.positions.numpy(), act.agents.yaws.numpy(), 7)
    hist_before = env._hist.clone()
    env.step(act, num_steps_to_take=7)
    got = env._hist[:, -7:].numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    # the window slid by 7 frames and the new frames are available
    assert torch.equal(env._hist[:, :-7], hist_before[:, 7:])
    assert bool(env._hist_avail[:, -7:].all())
    # speed = finite-difference velocity norm
    d = np.linalg.norm(want[:, 1:, :2] - want[:, :-1, :2], axis=-1) / 0.1
    np.testing.assert_allclose(env._hist_speed[:, -6:].numpy(), d, rtol=1e-4, atol=1e-5)
    assert env._frame_index == 7 and not env.is_done()
    env.step(random_action(B, 12, 1), num_steps_to_take=7)
    assert not env.is_done()
    env.step(random_action(B, 12, 2), num_steps_to_take=7)      # 14 + 7 >= 20
    assert env.is_done()
    with pytest.raises(RuntimeError):
        env.step(act, num_steps_to_take=1)


def test_observation_matches_per_agent_oracle():
    env = make_env()
    env.reset(scene_indices=[2, 0])
    B0 = env.current_num_agents
    env.step(random_action(B0, 10, 5), num_steps_to_take=4)       # make headings differ from the histories'
    obs = env.get_observation()["agents"]
    hist = env._hist.numpy().astype(np.float64)
    pose = hist[:, -1]
    sidx = env.current_agent_scene_index
    B, Th = hist.shape[:2]
    assert B == 4 + 5 and Th == 31
    assert obs["scene_index"].tolist() == [2] * 4 + [0] * 5
    for b in range(B):
        local, dyaw, wfa, afw = eo.agent_frame(hist[b], pose[b])
        av = env._hist_avail[b].numpy()
        np.testing.assert_allclose(obs["history_positions"][b].numpy()[av], local[av], atol=1e-4)
        np.testing.assert_allclose(obs["history_yaws"][b, :, 0].numpy()[av], dyaw[av], atol=1e-5)
        assert np.isnan(obs["history_positions"][b].numpy()[~av]).all()
        np.testing.assert_allclose(obs["world_from_agent"][b].numpy(), wfa, atol=1e-5)
        np.testing.assert_allclose(obs["agent_from_world"][b].numpy(), afw, atol=1e-4)
        # neighbours: same set, nearest first, padded slots unavailable
        want = eo.neighbors(pose, sidx, b, env.neighbor_radius, env.max_neighbors)
        nav = obs["all_other_agents_history_availabilities"][b].numpy()
        assert int(nav.any(-1).sum()) == len(want)
        for slot, j in enumerate(want):
            nl, ny, _, _ = eo.agent_frame(hist[j], pose[b])
            a = env._hist_avail[j].numpy()
            assert (nav[slot] == a).all()
            np.testing.assert_allclose(obs["all_other_agents_history_positions"][b, slot].numpy()[a], nl[a], atol=1e-4)
            np.testing.assert_allclose(obs["all_other_agents_history_yaws"][b, slot, :, 0].numpy()[a], ny[a], atol=1e-5)
            np.testing.assert_allclose(obs["all_other_agents_extents"][b, slot].numpy(), [0.8, 0.8, 1.75])
        for slot in range(len(want), nav.shape[0]):
            assert not nav[slot].any()
    assert float(obs["curr_speed"].min()) > 0
    # raster_from_agent from pixel size / ego centre (trajdata_utils.py:196-200)
    np.testing.assert_allclose(obs["raster_from_agent"][0].numpy(), [[2, 0, 8], [0, 2, 16], [0, 0, 1]])


def test_raster_matches_pixel_loop():
    env = make_env(area=12.0)
    env.reset(scene_indices=[1])
    obs = env.get_observation()["agents"]
    C, H, W = env.cfg.image_shape
    boxes = env._boxes[0].numpy().astype(np.float64)
    rfa = obs["raster_from_agent"][0].numpy().astype(np.float64)
    pose = env._hist[:, -1].numpy().astype(np.float64)
    mism = 0
    for b in range(pose.shape[0]):
        want = eo.rasterize(pose[b], boxes, rfa, H, W)
        got = obs["drivable_map"][b].numpy()
        mism += int((want != got).sum())          # fp32 vs fp64 may flip a pixel exactly on an edge
    assert mism <= 2
    assert not bool(obs["drivable_map"].all()) and bool(obs["drivable_map"].any())
    assert torch.equal(obs["image"][:, 0] > 0, obs["drivable_map"])
    assert torch.equal(obs["image"][:, 1] > 0, ~obs["drivable_map"])


class ConstantVelocityModel:
    """Stands in for DiffuserTrafficModel on the CPU: plans the constant-velocity continuation."""
    device = torch.device("cpu")

    def __init__(self, T=52, dt=0.1):
        self.T, self.dt, self.calls, self.guidance = T, dt, [], None

    def eval(self):
        return self

    def set_guidance(self, cfg, example_batch=None):
        self.guidance = cfg
        assert "extent" in example_batch and "scene_index" in example_batch

    def clear_guidance(self):
        self.guidance = None

    def get_action(self, obs, step_index=None, **kw):
        self.calls.append(step_index)
        k = torch.arange(1, self.T + 1, dtype=torch.float32)[None, :, None] * self.dt
        pos = torch.cat([obs["curr_speed"][:, None, None] * k, torch.zeros(obs["curr_speed"].shape[0], self.T, 1)], -1)
        return Action(pos, torch.zeros(pos.shape[0], self.T, 1)), {"action_samples": {}}


def test_guided_rollout_closed_loop():
    env = make_env(metrics={"coll": CollisionRate(), "offroad": OffRoadRate()})
    model = ConstantVelocityModel()
    policy = RolloutWrapper(agents_policy=model)
    guidance = [[{"name": "agent_collision", "weight": 1.0, "params": {"num_disks": 1, "buffer_dist": 0.2},
                  "agents": None}]] * 2
    stats, info = se.guided_rollout(env, policy, model, n_step_action=5, guidance_config=guidance,
                                    scene_indices=[0, 2], horizon=20)
    assert model.calls == [0, 5, 10, 15]
    assert model.guidance is None                      # cleared at the end
    assert info["scene_index"] == ["scene_000000", "scene_000002"]
    # the per-guidance success metrics ride along under the reference's key names (guidance_metrics.py:496-510)
    guide = {"guide_%s_s%dg0" % (m, s) for m in ("agent_collision", "agent_collision_disk", "social_dist")
             for s in (0, 1)}
    assert set(stats) == {"ade", "fde", "coll", "offroad"} | guide
    for k, v in stats.items():
        assert v.shape == (2,)
        if k in guide:                                                 # NaN but for the metric's own scene
            own = int(k.split("_s")[-1][0])
            assert np.isnan(v[1 - own]) and 0.0 <= v[own] <= 1.0
    assert set(env._metrics) == {"coll", "offroad"}                    # removed again after the rollout
    # the constant-velocity plan reproduces the recorded future
    assert float(stats["ade"].max()) < 1e-3 and float(stats["fde"].max()) < 1e-3
    assert env.is_done()


def test_heuristic_guidance_and_merge():
    env = make_env()
    heur = [
        {"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
        {"name": "map_collision", "weight": 10.0, "params": {"num_points_lw": [10, 10]}},
        {"name": "target_pos_at_time", "weight": 30000.0, "params": {"target_time": 40}},
        {"name": "global_target_pos", "weight": 10000.0,
         "params": {"target_time": 50, "urgency": 0.7, "pref_speed": 1.25, "min_progress_dist": 0.5, "dt": 0.1}},
    ]
    keep = json.dumps(heur)
    cfgs = se.compute_heuristic_guidance(heur, env, [0, 1], None)
    assert json.dumps(heur) == keep                                    # caller's config untouched
    assert len(cfgs) == 2 and [len(c) for c in cfgs] == [4, 4]
    for si, a in enumerate([5, 3]):
        names = [g["name"] for g in cfgs[si]]
        assert names == ["agent_collision", "map_collision", "target_pos_at_time", "global_target_pos"]
        for g in cfgs[si]:
            assert set(g) == {"name", "weight", "params", "agents"}
        tp = cfgs[si][2]
        assert tp["agents"] == list(range(a)) and tp["params"]["target_time"] == [39] * a
        # constant-velocity future: the local target sits speed * 4.0 s straight ahead
        scene = env._current_scenes[si]
        speed = np.linalg.norm(scene.get_agents_curr()[:, 2:4], axis=-1)
        np.testing.assert_allclose(np.array(tp["params"]["target_pos"]), np.stack([speed * 4.0, 0 * speed], -1), atol=1e-5)
        gt = cfgs[si][3]
        assert "target_time" not in gt["params"] and gt["params"]["min_progress_dist"] == 0.5
        assert gt["params"]["urgency"] == [0.7] * a and len(gt["params"]["target_pos"]) == a
    merged = se.merge_guidance_configs(cfgs, [[{"name": "x"}], []])
    assert len(merged[0]) == 5 and len(merged[1]) == 4 and len(cfgs[0]) == 4
    assert se.merge_guidance_configs(None, cfgs) is cfgs and se.merge_guidance_configs(cfgs, []) is cfgs
    # social grouping: two agents walking side by side form one group
    class Pair:
        dataset = env.dataset
        def get_agents_curr(self):
            return np.array([[0, 0, 1.2, 0, 0, 0, 0], [0, 0.8, 1.2, 0, 0, 0, 0], [30, 30, -1.0, 0, 0, 0, 3.14]], float)
    g = se.heuristic_social_group(Pair(), 0.1, group_dist_thresh=2.0, social_dist=1.5, cohesion=0.8)
    assert len(g) == 1 and g[0]["agents"] == [0, 1] and g[0]["params"]["leader_idx"] in (0, 1)


def test_config_dict_semantics():
    d = Dict()
    assert "a" not in d and d.a is not d.a and "a" not in d          # reads do not create keys
    d.a.b.c = 1
    assert d.to_dict() == {"a": {"b": {"c": 1}}}
    d.update({"a": {"b": {"e": 2}}, "f": [1, {"g": 3}]})
    assert d.a.b.c == 1 and d.a.b.e == 2 and d.f[1].g == 3
    d.lock()
    with pytest.raises(KeyError):
        d.zzz
    with pytest.raises(KeyError):
        d.a.new = 1
    d.a.b.c = 5
    e = d.clone()
    e.unlock()
    e.a.b.c = 6
    assert d.a.b.c == 5 and json.loads(d.dump())["a"]["b"]["c"] == 5


def test_scene_editing_config_defaults_and_cli_postprocessing():
    cfg = finalize_eval_config(SceneEditingConfig(), results_root_dir="out", prefix="p_", seed=3,
                               editing_source=["config", "heuristic"])
    assert cfg.eval_class == "Diffuser" and cfg.name == "p_scene_edit_eval" and cfg.seed == 3
    assert cfg.results_dir == os.path.join("out", "p_scene_edit_eval")
    assert cfg.n_step_action == 50 and cfg.num_simulation_steps == 50 and "trajdata" not in cfg
    assert cfg.policy.guide_clean is True and cfg.policy.class_free_guide_w == 0.0
    assert [h["name"] for h in cfg.edits.heuristic_config] == ["agent_collision", "map_collision", "target_pos_at_time"]
    with pytest.raises(KeyError):
        cfg.not_a_key


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason="reference configs only exist in the build container")
def test_reference_eval_jsons_load_unchanged():
    n = 0
    for root, _, files in os.walk(REF_CFG_DIR):
        for f in files:
            if not f.endswith(".json"):
                continue
            cfg = finalize_eval_config(SceneEditingConfig(), config_file=os.path.join(root, f))
            raw = json.load(open(os.path.join(root, f)))
            assert cfg.policy.num_action_samples == raw["policy"]["num_action_samples"]
            assert cfg.n_step_action == raw["trajdata"]["n_step_action"]
            assert cfg.num_simulation_steps == raw["trajdata"]["num_simulation_steps"]
            for h, r in zip(cfg.edits.heuristic_config, raw["edits"]["heuristic_config"]):
                assert h["name"] == r["name"] and h["weight"] == r["weight"]
            n += 1
    assert n >= 5


def test_reference_import_paths_resolve_through_the_shim():
    """A driver written against the reference's import paths (scripts/scene_editor.py:12-18,48) finds
    the MI355X implementation when trace_amd/compat is on the path (option B)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib\n"
            "from tbsim.configs.scene_edit_config import SceneEditingConfig\n"
            "from tbsim.utils.scene_edit_utils import guided_rollout, compute_heuristic_guidance, merge_guidance_configs\n"
            "from tbsim.policies.wrappers import RolloutWrapper, PolicyWrapper\n"
            "from tbsim.policies.common import Action, RolloutAction\n"
            "from tbsim.envs.env_trajdata import EnvUnifiedSimulation\n"
            "cfg = SceneEditingConfig()\n"
            "cls = getattr(importlib.import_module('tbsim.evaluation.policy_composers'), cfg.eval_class)\n"
            "import trace_amd.policy, trace_amd.scene_edit\n"
            "assert cls is trace_amd.policy.Diffuser and guided_rollout is trace_amd.scene_edit.guided_rollout\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "trace_amd", "compat") + os.pathsep + root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_gt_replay_policy_reproduces_the_recorded_future():
    """GTNaNPolicy (hardcoded.py:8-29) on the env's target_positions: zero ADE / FDE in closed loop, and
    the targets are expressed in the CURRENT agent frame after a step."""
    from trace_amd.policy import GTNaNPolicy
    env = make_env()
    policy = RolloutWrapper(agents_policy=GTNaNPolicy("cpu"))
    stats, _ = se.guided_rollout(env, policy, None, n_step_action=4, guidance_config=None,
                                 scene_indices=[0, 1, 2], horizon=20, use_gt=True)
    assert float(stats["ade"].max()) < 1e-4 and float(stats["fde"].max()) < 1e-4
    env.reset(scene_indices=[1])
    obs = env.get_observation()["agents"]
    assert obs["target_positions"].shape == (3, 52, 2) and bool(obs["target_availabilities"].all())
    speed = obs["curr_speed"]
    np.testing.assert_allclose(obs["target_positions"][:, 9, 0].numpy(), (speed * 1.0).numpy(), atol=1e-5)
    np.testing.assert_allclose(obs["target_positions"][:, :, 1].numpy(), 0.0, atol=1e-5)
