"""Host-side logic of the product (no GPU, no kernels): schedule tables, parameter flattening,
guidance-config parsing / layer packing / sample selection, scene sharding.  The selection logic is
checked against the oracle's restatement of choose_action_from_guidance (guidance_loss.py:15-41)."""
import os

import numpy as np
import pytest
import torch

from oracle import trace_oracle as orc
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.distributed import shard_scenes, slice_batch, slice_guidance
from trace_amd.guidance import (DiffuserGuidance, GuidanceConfig, GuidancePlan,
                                verify_guidance_config_list)
from trace_amd.schedule import Schedule
from trace_amd.unet_params import flatten_unet_params
from trace_amd.weights import extract_policy_state_dict, random_state_dict, state_dict_shapes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_schedule_tables_match_reference_golden_bitwise():
    comp = np.load(os.path.join(GOLDEN, "components.npz"))
    s = Schedule(100)
    for k in ("betas", "alphas_cumprod", "posterior_variance") + Schedule.NAMES:
        assert np.array_equal(comp["tab/" + k], getattr(s, k).numpy()), k
    c = s.coefs(99)
    assert c["t"] == 99 and abs(c["sqrt_recip_alphas_cumprod"] - 2029.2) < 0.1


def test_flatten_unet_params_is_a_permutation_of_the_denoiser_weights():
    cfg = ModelConfig()
    sd = random_state_dict(cfg, 0)
    flat = flatten_unet_params(cfg, sd)
    assert flat.dtype == torch.float32 and flat.dim() == 1 and not flat.is_cuda
    unet = {k: v for k, v in sd.items() if k.startswith("model.")}
    assert flat.numel() == sum(v.numel() for v in unet.values())
    assert abs(float(flat.double().sum()) - sum(float(v.double().sum()) for v in unet.values())) < 1e-3
    # shapes table and the generated dict agree; prefixes of a PL checkpoint are stripped
    shapes = state_dict_shapes(cfg)
    assert set(shapes) == set(sd) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    ck = {"nets.policy." + k: v for k, v in sd.items()}
    ck.update({"ema_policy." + k: v + 1 for k, v in sd.items()})
    got = extract_policy_state_dict(ck, use_ema=True)
    k0 = next(iter(sd))
    assert set(got) == set(sd) and torch.equal(got[k0], sd[k0] + 1)
    assert torch.equal(extract_policy_state_dict(ck, use_ema=False)[k0], sd[k0])


def test_guidance_config_checks_mirror_the_reference():
    ok = {"name": "target_pos", "weight": 1.0, "params": {"target_pos": [[0.0, 0.0]]}, "agents": None}
    assert GuidanceConfig.from_dict(ok).name == "target_pos"
    with pytest.raises(AssertionError):                      # guidance_loss.py:88-90: exact key set
        GuidanceConfig.from_dict({"name": "target_pos", "weight": 1.0, "params": {}})
    with pytest.raises(AssertionError):                      # :79 unknown name
        GuidanceConfig.from_dict(dict(ok, name="nope"))
    with pytest.raises(NotImplementedError):
        GuidanceConfig.from_dict(dict(ok, name="amp_value"))
    with pytest.raises(AssertionError):
        verify_guidance_config_list([])
    assert verify_guidance_config_list([[], [ok]]) and not verify_guidance_config_list([[], []])
    with pytest.raises(AssertionError):
        DiffuserGuidance([])


def _plan(aps, N, configs, seed=3):
    cfg = ModelConfig()
    batch = syn.make_batch(cfg, aps, seed=seed, extent=(0.9, 0.7, 1.75), dense=True)
    g = DiffuserGuidance(configs, batch)
    return batch, GuidancePlan(g, batch, N, cfg.horizon, "cpu")


def test_plan_merges_disjoint_configs_into_layers_and_normalises_by_mask():
    aps, N = [3, 2, 4], 5
    cfg = ModelConfig()
    batch = syn.make_batch(cfg, aps, seed=3)
    tp = syn.target_pos_guidance(batch, aps, weight=30000.0, seed=1)
    ac = [[{"name": "agent_collision", "weight": 1000.0, "params": {"num_disks": 2, "buffer_dist": 0.2},
            "agents": None}] for _ in aps]
    ac[1] = []
    ac[2][0] = dict(ac[2][0], agents=[0, 2])
    configs = syn.merge_guidance(tp, ac)
    _, plan = _plan(aps, N, configs)
    kinds = sorted(ly["kind"] for ly in plan.layers)
    assert kinds == ["agent_collision", "target_pos"]           # 5 configs -> 2 launches per step
    tl = [ly for ly in plan.layers if ly["kind"] == "target_pos"][0]
    assert bool(tl["mask"].all())
    # weight / (|mask| * N), guidance_loss.py:919 (mean over masked agents x samples)
    sc = tl["tables"]["scale"]
    exp = torch.tensor([30000.0 / (a * N) for a in aps for _ in range(a)])
    assert torch.allclose(sc, exp)
    al = [ly for ly in plan.layers if ly["kind"] == "agent_collision"][0]
    assert al["mask"].tolist() == [True] * 3 + [False] * 2 + [True, False, True, False]
    assert [i.key_name for i in plan.instances] == [
        "target_pos_scene_000_00", "agent_collision_scene_000_01", "target_pos_scene_001_00",
        "target_pos_scene_002_00", "agent_collision_scene_002_01"]


def test_scene_coupled_configs_of_one_scene_never_share_a_launch():
    """Two agent_collision configs with equal parameters and disjoint agents in ONE scene stay separate layers
    (the pair kernel doubles a pair's gradient inside one config only, guidance_loss.py:272-282); configs of
    different scenes still merge.  Agents a layer does not cover carry scale -1, covered ones weight/(|mask| N)
    even when the weight is 0."""
    aps, N = [3, 2], 4
    ac = lambda w, agents: {"name": "agent_collision", "weight": w, "params": {"num_disks": 2, "buffer_dist": 0.2},
                            "agents": agents}
    configs = [[ac(1.0, [0, 1]), ac(0.0, [2])], [ac(3.0, None)]]
    _, plan = _plan(aps, N, configs)
    layers = [ly for ly in plan.layers if ly["kind"] == "agent_collision"]
    assert len(layers) == 2
    assert layers[0]["mask"].tolist() == [True, True, False, True, True]
    assert layers[1]["mask"].tolist() == [False, False, True, False, False]
    assert layers[1]["tables"]["scale"].tolist() == [-1.0, -1.0, 0.0, -1.0, -1.0]
    assert torch.allclose(layers[0]["tables"]["scale"], torch.tensor([1 / 8, 1 / 8, -1.0, 3 / 8, 3 / 8]))


def test_target_speed_list_form_raises_like_the_reference_would():
    aps = [2]
    bad = [[{"name": "target_speed", "weight": 1.0,
             "params": {"target_speed": [1.0, 2.0], "dt": 0.1, "mode": "use_action"}, "agents": None}]]
    with pytest.raises(ValueError):
        _plan(aps, 2, bad)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_choose_action_matches_oracle_selection(seed):
    aps, N = [3, 2, 4, 1], 6
    cfg = ModelConfig()
    batch = syn.make_batch(cfg, aps, seed=seed)
    tp = syn.target_pos_guidance(batch, aps, weight=1.0, seed=seed)
    extra = [[] for _ in aps]
    extra[0] = [{"name": "agent_collision", "weight": 1.0, "params": {"num_disks": 1}, "agents": None}]
    extra[1] = [{"name": "min_speed", "weight": 1.0, "params": {"min_speed": 1.0, "dt": 0.1},
                 "agents": [1]}]                               # partial cover -> agent 0 of scene 1 invalid
    extra[2] = [{"name": "map_collision", "weight": 1.0, "params": {"num_points_lw": (4, 4)},
                 "agents": None}]
    # an unguided scene keeps sample 0 -- and, in the reference, resets every scene before it
    # (zero-width loss slice => all agents "valid", guidance_loss.py:26-39): reproduced
    empty_scene = {0: None, 1: 1, 2: 3}[seed]
    if empty_scene is not None:
        tp[empty_scene], extra[empty_scene] = [], []
    configs = syn.merge_guidance(tp, extra)
    g = DiffuserGuidance(configs, batch)
    plan = GuidancePlan(g, batch, N, cfg.horizon, "cpu")
    gen = torch.Generator().manual_seed(seed)
    B = sum(aps)
    layer_losses = [torch.rand(B * N, generator=gen) * ly["mask"].repeat_interleave(N) for ly in plan.layers]
    losses = plan.guide_losses(layer_losses)
    names = [[c["name"] for c in sc] for sc in configs]
    ref = orc.choose_action_from_guidance(B, names, losses)
    got = plan.choose_action(layer_losses)
    assert torch.equal(got, ref), (got, ref)
    # the public free function on the per-guide NaN-padded losses (guidance_loss.py:15-41), objects or dicts
    from trace_amd.guidance import choose_action_from_guidance
    preds = {"positions": torch.zeros(B, N, cfg.horizon, 2)}
    assert torch.equal(choose_action_from_guidance(preds, batch, g.guide_configs, losses), ref)
    assert torch.equal(choose_action_from_guidance(preds, batch, configs, losses), ref)
    # NaN outside each guide's mask (guidance_loss.py:916-918)
    if empty_scene != 1:
        v = losses["min_speed_scene_001_01"]
        assert bool(torch.isnan(v[3]).all()) and not bool(torch.isnan(v[4]).any())
        assert int(got[3]) == 0                                 # not covered by every guide of its scene
    if empty_scene is None:
        assert len(set(got[:3].tolist())) == 1                  # scene-level choice (agent_collision)


def test_shard_scenes_is_a_balanced_contiguous_partition():
    for aps in ([32] * 8, [5, 1, 1, 1, 20, 3], [7], [1, 1], [128]):
        for ws in (1, 2, 3, 4, 8):
            r = shard_scenes(aps, ws)
            assert len(r) == ws and r[0][0] == 0 and r[-1][1] == len(aps)
            assert all(r[i][1] == r[i + 1][0] and r[i][0] <= r[i][1] for i in range(ws - 1))
    r = shard_scenes([32] * 8, 4)
    assert r == [(0, 2), (2, 4), (4, 6), (6, 8)]
    aps = [5, 1, 1, 1, 20, 3]
    loads = [sum(aps[b:e]) for b, e in shard_scenes(aps, 2)]
    assert max(loads) <= 23                                     # the 20-agent scene is not split


def test_slice_batch_keeps_scenes_whole():
    cfg = ModelConfig()
    aps = [3, 2, 4]
    batch = syn.make_batch(cfg, aps, seed=0)
    sub, (a0, a1) = slice_batch(batch, aps, (1, 3))
    assert (a0, a1) == (3, 9) and sub["history_positions"].shape[0] == 6
    assert sub["scene_index"].tolist() == batch["scene_index"][3:9].tolist()
    g = [[1], [2], [3]]
    assert slice_guidance(g, (1, 3)) == [[2], [3]] and slice_guidance(None, (0, 1)) is None


def test_compat_package_exposes_the_reference_import_paths():
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    compat = os.path.join(root, "trace_amd", "compat")
    assert not any(m == "tbsim" or m.startswith("tbsim.") for m in sys.modules), "reference leaked into tests"
    sys.path.insert(0, compat)
    try:
        from trace_amd import policy
        pc = importlib.import_module("tbsim.evaluation.policy_composers")   # scene_editor.py:48
        assert getattr(pc, "Diffuser") is policy.Diffuser
        w = importlib.import_module("tbsim.policies.wrappers")
        assert w.PolicyWrapper is policy.PolicyWrapper and w.RolloutWrapper is policy.RolloutWrapper
        c = importlib.import_module("tbsim.policies.common")
        assert c.Action is policy.Action and c.RolloutAction is policy.RolloutAction
        assert importlib.import_module("tbsim.algos.algos").DiffuserTrafficModel is policy.DiffuserTrafficModel
        gl = importlib.import_module("tbsim.utils.guidance_loss")
        assert gl.DiffuserGuidance is DiffuserGuidance
        from trace_amd import guidance as tg
        assert gl.choose_action_from_guidance is tg.choose_action_from_guidance
    finally:
        sys.path.remove(compat)
        for m in [m for m in sys.modules if m == "tbsim" or m.startswith("tbsim.")]:
            del sys.modules[m]


def test_get_checkpoint_layout(tmp_path):
    """Checkpoint discovery as the reference's get_checkpoint (experiment_utils.py:39-80) and the
    Lightning state_dict prefixes (algos.py:54,96)."""
    import json
    import torch
    from trace_amd.config import ModelConfig
    from trace_amd.policy import get_checkpoint, modality_image_shape
    from trace_amd.weights import extract_policy_state_dict, random_state_dict

    cfg = ModelConfig()
    sd = random_state_dict(cfg, 3)
    run = tmp_path / "trace" / "orca_mixed" / "run0"
    (run / "checkpoints").mkdir(parents=True)
    ck = {"epoch": 1, "state_dict": {**{"ema_policy." + k: v for k, v in sd.items()},
                                     **{"nets.policy." + k: v * 2 for k, v in sd.items()}}}
    torch.save(ck, run / "checkpoints" / "iter40000_ep1_valLoss0.50.ckpt")
    torch.save(ck, run / "checkpoints" / "iter20000_ep0_valLoss0.70.ckpt")
    exp = {"algo": cfg.to_algo_dict(), "env": {"rasterizer": {"include_hist": False, "num_sem_layers": 7,
                                                              "raster_size": 224}}}
    (run / "config.json").write_text(json.dumps(exp))
    ckpt, cfgp = get_checkpoint("iter40000", str(tmp_path / "trace" / "orca_mixed"))
    assert ckpt.endswith("iter40000_ep1_valLoss0.50.ckpt") and cfgp.endswith("run0/config.json")
    with pytest.raises(FileNotFoundError):
        get_checkpoint("iter99999", str(tmp_path / "trace"))
    with pytest.raises(AssertionError):
        get_checkpoint("iter", str(tmp_path / "trace"))
    loaded = torch.load(ckpt, map_location="cpu", weights_only=True)["state_dict"]
    ema = extract_policy_state_dict(loaded, use_ema=True)
    raw = extract_policy_state_dict(loaded, use_ema=False)
    assert list(ema) == list(sd) and all(torch.equal(ema[k], sd[k]) for k in sd)
    assert all(torch.equal(raw[k], sd[k] * 2) for k in sd)
    mc = ModelConfig.from_algo_dict(json.loads((run / "config.json").read_text())["algo"],
                                    image_shape=modality_image_shape(exp))
    assert mc.image_shape == (7, 224, 224) and mc.unet_dims == cfg.unet_dims and mc.norm_div == cfg.norm_div
    assert modality_image_shape({"algo": {"history_num_frames": 30},
                                 "env": {"rasterizer": {"include_hist": True, "num_sem_layers": 2,
                                                        "raster_size": 100}}}) == (33, 100, 100)
