"""GPU tests of the callers either side of the sampler: the
device-resident environment feeding the HIP sampler in a replanning loop, driven through the
reference-compatible command line (scripts/scene_editor.py)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene_editor():
    spec = importlib.util.spec_from_file_location("scene_editor_cli", os.path.join(ROOT, "scripts", "scene_editor.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _eval_json(tmp_path, n_step_action, sim_steps, heuristics, samples=4, w=0.0):
    cfg = {"name": "closed_loop_test", "eval_class": "Diffuser", "seed": 1,
           "policy": {"num_action_samples": samples, "class_free_guide_w": w, "guide_as_filter_only": False,
                      "guide_with_gt": False, "guide_clean": True},
           "trajdata": {"n_step_action": n_step_action, "num_simulation_steps": sim_steps,
                        "num_scenes_to_evaluate": 4, "eval_scenes": [0, 1, 2, 3]},
           "edits": {"editing_source": ["heuristic"], "guidance_config": [], "heuristic_config": heuristics}}
    p = tmp_path / "eval.json"
    p.write_text(json.dumps(cfg))
    return str(p)


HEUR = [
    {"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
    {"name": "map_collision", "weight": 10.0, "params": {"num_points_lw": [10, 10]}},
    {"name": "global_target_pos", "weight": 10000.0,
     "params": {"target_time": 50, "urgency": 0.7, "pref_speed": 1.25, "min_progress_dist": 0.5, "dt": 0.1}},
]


def test_scene_editor_cli_closed_loop(tmp_path):
    """cfg5 in miniature: 4 scenes x 6 agents, replanning every 10 frames for 30 frames, combined
    target + collision guidance, two scenes per batch; writes the reference's stats.json."""
    se = _scene_editor()
    cfgf = _eval_json(tmp_path, 10, 30, HEUR)
    args = ["--config_file", cfgf, "--results_root_dir", str(tmp_path / "out"), "--num_scenes_per_batch", "2",
            "--allow_random_init", "--synthetic_scenes", "4", "--synthetic_agents", "6"]
    stats = se.main(args)
    assert stats["scene_index"].tolist() == ["scene_%06d" % i for i in range(4)]
    for k in ("ade", "fde", "all_collision_rate", "all_off_road_rate"):
        assert stats[k].shape == (4,) and np.isfinite(stats[k]).all(), k
    on_disk = json.load(open(tmp_path / "out" / "closed_loop_test" / "stats.json"))
    assert on_disk["ade"] == stats["ade"].tolist()
    # agents moved: a 3 s rollout of 1-2 m/s pedestrians under random weights stays within tens of metres
    assert 0.0 < float(stats["fde"].max()) < 60.0
    # --ground_truth replays the recorded future through the same loop (scene_editor.py:53-58)
    gt = se.main(args + ["--ground_truth", "--prefix", "gt_"])
    assert float(gt["ade"].max()) < 1e-3 and float(gt["fde"].max()) < 1e-3


def test_closed_loop_replan_matches_oracle_on_env_observation():
    """One replanning call on an observation produced by the device environment after a step: the HIP
    sampler and the CPU oracle get the same observation (incl. NaN-marked unavailable history), the
    same encoded map grid and the same noise tape."""
    from oracle import trace_oracle as orc
    from trace_amd import synthetic as syn
    from trace_amd.config import ModelConfig
    from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict
    from trace_amd.policy import Action, RolloutAction
    from trace_amd.sampler import TraceSampler
    from trace_amd.scene_edit import compute_heuristic_guidance
    from trace_amd.sim_env import SceneBatchEnv
    from trace_amd.weights import random_state_dict

    cfg = ModelConfig()
    dev = torch.device("cuda:0")
    sd = random_state_dict(cfg, 0)
    env = SceneBatchEnv(cfg, [5, 4], device=dev, seed=2, num_simulation_steps=40, area=14.0)
    env.reset(scene_indices=[0, 1])
    B = env.current_num_agents
    env._hist_avail[0, :25] = False                                    # one agent with a short observed history
    # walk a little so histories, headings and neighbour sets are not the initial ones
    k = torch.arange(1, 53, device=dev, dtype=torch.float32)[None, :, None] * 0.1
    pos = torch.cat([1.3 * k, 0.2 * k * k], -1).repeat(B, 1, 1)
    yaw = (0.3 * k).repeat(B, 1, 1)
    env.step(RolloutAction(agents=Action(pos, yaw), agents_info={}), num_steps_to_take=10)
    obs = dict(env.get_observation()["agents"])
    assert bool(torch.isnan(obs["history_positions"]).any())           # short histories are NaN-marked
    enc = MapEncoder(random_map_encoder_state_dict(2, cfg.map_grid_feature_dim, cfg.image_shape[1:], seed=0), dev)
    with torch.no_grad():
        obs["map_grid_feat"] = enc(obs["image"])[1]
        obs["map_grid_feat_non_cond"] = enc(torch.ones_like(obs["image"]) * cfg.conditioning_drop_fill)[1]
    heur = [{"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
            {"name": "target_pos", "weight": 30000.0, "params": {"target_time": 40}}]
    guidance = compute_heuristic_guidance(heur, env, [0, 1], None)     # (resets the env; obs was taken before)
    N, n_steps = 3, 12
    init, steps = syn.make_noise(cfg, B, N, seed=4)
    s = TraceSampler(cfg, sd, "cuda:0")
    s.set_guidance(guidance, obs)
    out = s(obs, num_samp=N, class_free_guide_w=0.5, guide_clean=True, noise_init=init, noise_steps=steps,
            n_steps=n_steps, return_guidance_losses=True)
    torch.cuda.synchronize()
    cpu_obs = {k_: (v.cpu() if torch.is_tensor(v) else v) for k_, v in obs.items()}
    o = orc.OracleSampler(cfg, sd)
    o.set_guidance(guidance)
    ref = o.sample(cpu_obs, N, 0.5, True, True, noise_init=init, noise_steps=steps, n_steps=n_steps,
                   return_diffusion=False)
    got = out["pred_traj_scaled"].cpu()
    assert bool(torch.isfinite(got).all())
    err = float(((got - ref["pred_traj"]).abs() / (1 + ref["pred_traj"].abs())).max())
    assert err < 1e-4, err


def test_composer_loads_lightning_checkpoint(tmp_path):
    """``Diffuser(eval_cfg, device).get_policy()`` on a checkpoint directory laid out like the
    reference's (``<ckpt_dir>/run/checkpoints/iter*.ckpt`` + ``config.json``; ``ema_policy.*`` keys):
    same plan as the model built directly from the state_dict."""
    from trace_amd import synthetic as syn
    from trace_amd.config import ModelConfig
    from trace_amd.eval_config import SceneEditingConfig, finalize_eval_config
    from trace_amd.policy import Diffuser, DiffuserTrafficModel, PolicyWrapper
    from trace_amd.weights import random_state_dict

    cfg = ModelConfig()
    sd = random_state_dict(cfg, 5)
    run = tmp_path / "ckpt" / "orca_mixed" / "run0"
    (run / "checkpoints").mkdir(parents=True)
    torch.save({"epoch": 3, "state_dict": {**{"ema_policy." + k: v for k, v in sd.items()},
                                           **{"nets.policy." + k: v + 1 for k, v in sd.items()}}},
               run / "checkpoints" / "iter40000_ep3.ckpt")
    (run / "config.json").write_text(json.dumps(
        {"algo": cfg.to_algo_dict(), "env": {"rasterizer": {"include_hist": False, "num_sem_layers": 2,
                                                            "raster_size": 224}}}))
    ev = finalize_eval_config(SceneEditingConfig(), policy_ckpt_dir=str(tmp_path / "ckpt" / "orca_mixed"),
                              policy_ckpt_key="iter40000")
    policy, exp_cfg = Diffuser(ev, "cuda:0").get_policy()
    assert exp_cfg["algo"]["horizon"] == 52
    aps = [3, 2]
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in syn.make_batch(cfg, aps, seed=2).items()}
    init, steps = syn.make_noise(cfg, sum(aps), 10, seed=1)
    a1, _ = policy.get_action(batch, step_index=0, noise_init=init, noise_steps=steps)
    direct = PolicyWrapper.wrap_controller(DiffuserTrafficModel(cfg, sd, "cuda:0"), num_action_samples=10,
                                           class_free_guide_w=0.0, guide_as_filter_only=False,
                                           guide_with_gt=False, guide_clean=True)
    a2, _ = direct.get_action(batch, step_index=0, noise_init=init, noise_steps=steps)
    assert torch.equal(a1.positions, a2.positions) and torch.equal(a1.yaws, a2.yaws)


def test_env_rasterize_kernel_vs_pixel_loop_and_torch():
    """trace_env_rasterize (HIP) against the per-pixel numpy oracle on a few agents and against the
    torch restatement on a full batch; then its HBM write rate is printed for the record."""
    from oracle import env_oracle as eo
    from trace_amd.config import ModelConfig
    from trace_amd.sim_env import SceneBatchEnv
    cfg = ModelConfig()
    env = SceneBatchEnv(cfg, [6, 5, 7], device="cuda:0", seed=5, area=16.0)
    env.reset(scene_indices=[0, 1, 2])
    pose = env._hist[:, -1]
    img, drv = env._rasterize(pose)
    ref = env._rasterize_torch(pose)
    C, H, W = cfg.image_shape
    assert img.shape == (18, C, H, W) and drv.shape == (18, H, W) and drv.dtype == torch.bool
    assert int((drv != ref).sum()) <= 4                      # a pixel exactly on a box edge may flip
    assert torch.equal(img[:, 0] > 0, drv) and torch.equal(img[:, 1] > 0, ~drv)
    assert 0.02 < float((~drv).float().mean()) < 0.9
    rfa = env._rfa.cpu().numpy().astype(np.float64)
    for b in (0, 7, 17):
        want = eo.rasterize(pose[b].cpu().numpy().astype(np.float64),
                            env._boxes[int(env._scene_index[b])].cpu().numpy().astype(np.float64), rfa, H, W)
        assert int((want != drv[b].cpu().numpy()).sum()) <= 3, b
    # bandwidth: 1024 agents -> (2 * 4 + 1) bytes per pixel written
    big = SceneBatchEnv(cfg, [32] * 32, device="cuda:0", seed=1)
    big.reset()
    p = big._hist[:, -1]
    big._rasterize(p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        big._rasterize(p)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = p.shape[0] * H * W * (4 * C + 1) / 1e9
    print("env_rasterize: %.3f ms for %d agents, %.0f GB/s written" % (ms, p.shape[0], gb / (ms * 1e-3)))


def test_nuscenes_shaped_replan_with_seven_channel_raster():
    """The nuScenes-shaped variant of the closed loop (configs/eval/nusc/target_pos.json:175-188 of the reference: a
    7-layer semantic raster): the environment rasterises 7 channels, the checkpoint carries a 7-channel stem, and one
    guided replanning call goes image -> HIP map encoder -> sampler.  Checked against the same call with the map
    encoder on PyTorch-ROCm ops (the encoder's own checker, tests/test_gpu_map_encoder.py) and for determinism."""
    import dataclasses
    from trace_amd import synthetic as syn
    from trace_amd.config import ModelConfig
    from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict
    from trace_amd.policy import DiffuserTrafficModel, PolicyWrapper
    from trace_amd.scene_edit import compute_heuristic_guidance
    from trace_amd.sim_env import SceneBatchEnv
    from trace_amd.weights import random_state_dict

    cfg = dataclasses.replace(ModelConfig(), image_shape=(7, 224, 224))
    dev = torch.device("cuda:0")
    sd = dict(random_state_dict(cfg, 0))
    sd.update(random_map_encoder_state_dict(7, cfg.map_grid_feature_dim, (224, 224), seed=3))
    env = SceneBatchEnv(cfg, [6, 5], device=dev, seed=5, num_simulation_steps=40, area=14.0)
    env.reset(scene_indices=[0, 1])
    B = env.current_num_agents
    heur = [{"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 1, "buffer_dist": 0.2}},
            {"name": "target_pos", "weight": 30000.0, "params": {"target_time": 40}}]
    guidance = compute_heuristic_guidance(heur, env, [0, 1], None)
    env.reset(scene_indices=[0, 1])
    obs = dict(env.get_observation()["agents"])
    assert obs["image"].shape == (B, 7, 224, 224)
    N, n_steps = 4, 10
    init, steps = syn.make_noise(cfg, B, N, seed=9)
    outs = []
    for backend in ("hip", "hip", "torch"):
        model = DiffuserTrafficModel(cfg, sd, dev, map_encoder=MapEncoder(sd, dev, backend=backend))
        assert model.sampler.map_encoder.backend == backend
        model.set_guidance(guidance, obs)
        out = model.sampler(obs, num_samp=N, class_free_guide_w=0.5, guide_clean=True, noise_init=init,
                            noise_steps=steps, n_steps=n_steps)
        outs.append(out["pred_traj_scaled"].cpu())
    assert bool(torch.isfinite(outs[0]).all())
    assert torch.equal(outs[0], outs[1])                                 # same bits from two fresh models
    err = float(((outs[0] - outs[2]).abs() / (1 + outs[2].abs())).max())
    assert err < 2e-4, err                                              # encoder round-off (5e-6) through 10 guided steps
