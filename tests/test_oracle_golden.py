"""Pins the CPU oracle (oracle/trace_oracle.py) against golden vectors emitted by the
reference itself (oracle/gen_golden.py).  CPU only.  Every comparison is BIT-EXACT (rtol = atol = 0):
the oracle runs the same ATen ops in the same order on the same torch build as the reference did."""
import os

import numpy as np
import pytest
import torch

from oracle import trace_oracle as orc
from oracle.gen_golden import CASES, FULL_DIFFUSION, case_inputs, loss_component_configs
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.weights import random_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def _close(a, b, rtol=0, atol=0):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


@pytest.fixture(scope="module")
def comp():
    return _load("components.npz")


@pytest.fixture(scope="module")
def setup():
    cfg = ModelConfig()
    sd = random_state_dict(cfg, 0)
    aps = [3, 2]
    batch = syn.make_batch(cfg, aps, seed=1, extent=(0.9, 0.7, 1.75), dense=True)
    return cfg, sd, aps, batch


def test_schedule_tables_bit_exact(comp):
    tab = orc.schedule_tables(100)
    for k, v in tab.items():
        assert np.array_equal(comp["tab/" + k], v.numpy()), k


def test_conditioning(comp, setup):
    cfg, sd, aps, batch = setup
    aux = orc.get_aux_info(sd, cfg, batch, True)
    _close(aux["cond_feat"], comp["aux/cond_feat"], 0, 0)
    _close(aux["non_cond_feat"], comp["aux/non_cond_feat"], 0, 0)


def test_dynamics_forward_and_grad(comp, setup):
    cfg, sd, aps, batch = setup
    g = torch.Generator().manual_seed(11)
    B = sum(aps)
    act = (torch.randn(B, cfg.horizon, 2, generator=g) * torch.tensor([6.0, 3.0])).requires_grad_()
    init = torch.zeros(B, 4)
    init[:, 2] = batch["curr_speed"] * torch.tensor([1.0, 30.0, -11.0, 12.0, 0.02])
    st = orc.unicycle_rollout(cfg, init, act)
    wgt = torch.randn(st.shape, generator=g)
    gr, = torch.autograd.grad((st * wgt).sum(), act)
    _close(st.detach(), comp["dyn/states"], 0, 0)
    _close(gr, comp["dyn/grad"], 0, 0)
    # the fixture must exercise every clip
    a = act.detach()
    assert (a[..., 0] > cfg.acce_bound[1]).any() and (a[..., 0] < cfg.acce_bound[0]).any()
    assert (st[..., 2] >= cfg.vbound[1]).any() and (st[..., 2] <= cfg.vbound[0]).any()
    assert (gr[..., 1] == 0).any() and (gr[..., 1] != 0).any()


def test_map_query_and_unet(comp, setup):
    cfg, sd, aps, batch = setup
    g = torch.Generator().manual_seed(11)
    B = sum(aps)
    torch.randn(B, cfg.horizon, 2, generator=g)
    torch.randn(B, cfg.horizon, 4, generator=g)
    x = torch.randn(B, cfg.horizon, 6, generator=g)
    aux = orc.get_aux_info(sd, cfg, batch, True)
    feats = orc.query_map_feats(cfg, x, aux["map_grid_feat"], aux["raster_from_agent"])
    _close(feats, comp["mapq/feats"], 0, 0)
    xin = torch.cat([x, feats], dim=-1).requires_grad_()
    t = torch.tensor([99, 50, 0, 7, 63])
    y = orc.unet_forward(sd, cfg, xin, aux["cond_feat"], t)
    wy = torch.randn(y.shape, generator=g)
    gx, = torch.autograd.grad((y * wy).sum(), xin)
    _close(y.detach(), comp["unet/out"], 0, 0)
    _close(gx, comp["unet/grad_in"], 0, 0)


def test_guidance_objectives(comp, setup):
    cfg, sd, aps, batch = setup
    traj0 = torch.tensor(comp["loss/traj"])
    active = {}
    for lname, cfgs in loss_component_configs(batch, aps).items():
        tr = traj0.clone().requires_grad_()
        guid = orc.Guidance(cfgs)
        guid.update(global_t=7)
        tot, per = guid.loss(tr, batch)
        gt, = torch.autograd.grad(tot, tr)
        _close(tot.detach(), comp["loss/%s/total" % lname], 0, 0)
        _close(gt, comp["loss/%s/grad" % lname], 0, 0)
        for k, v in per.items():
            _close(v, comp["loss/%s/per/%s" % (lname, k)], 0, 0)
        active[lname] = float(gt.abs().max())
    # every objective must be active in the fixture (non-trivial gradient)
    for k, v in active.items():
        assert v > 0, k


@pytest.mark.parametrize("name", CASES)
def test_sampler_end_to_end(name):
    c = case_inputs(name)
    gold = _load("sample_%s.npz" % name)
    o = orc.OracleSampler(c["cfg"], random_state_dict(c["cfg"], c["sd_seed"]))
    o.set_guidance(c["guidance"])
    res = o.get_action(c["batch"], c["num_samp"], c["w"], guide_as_filter_only=not c["apply"],
                       guide_clean=c["clean"], step_index=c["step_index"],
                       noise_init=c["noise_init"], noise_steps=c["noise_steps"])
    s = res["samples"]
    if name in FULL_DIFFUSION:
        _close(s["diffusion_steps"], gold["diffusion_steps"], 0, 0)
    else:
        _close(s["diffusion_steps"][:, :, :, -1], gold["diffusion_steps_last"], 0, 0)
    _close(s["positions"], gold["positions"], 0, 0)
    _close(s["yaws"], gold["yaws"], 0, 0)
    if c["guidance"] is not None:
        for k, v in s["guide_losses"].items():
            _close(v, gold["guide_loss/" + k], 0, 0)
        assert np.array_equal(res["act_idx"].numpy(), gold["act_idx"])


def test_social_group_vs_reference_golden(setup):
    """SocialGroupLoss (guidance_loss.py:744-814) with the reference's own torch.randint / torch.rand
    draws injected (tests/golden/social_group.npz, written by oracle/gen_golden_social.py)."""
    from oracle.gen_golden_social import social_configs
    cfg, sd, aps, batch = setup
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "social_group.npz")))
    traj0 = torch.tensor(dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "components.npz")))["loss/traj"])
    for name, cfgs in social_configs().items():
        guid = orc.Guidance(cfgs)
        for si, cur in enumerate(guid.scenes):
            for gi, (gname, _, _, fn) in enumerate(cur):
                key = "%s_scene_%03d_%02d" % (gname, si, gi)
                fn.draw = (torch.tensor(gold["%s/randint/%s" % (name, key)]),
                           torch.tensor(gold["%s/rand/%s" % (name, key)]))
        tr = traj0.clone().requires_grad_()
        tot, per = guid.loss(tr, batch)
        gt, = torch.autograd.grad(tot, tr)
        assert torch.equal(tot.detach(), torch.tensor(gold[name + "/total"])), name
        assert torch.equal(gt, torch.tensor(gold[name + "/grad"])), name
        for k, v in per.items():
            assert torch.equal(torch.nan_to_num(v), torch.nan_to_num(torch.tensor(gold["%s/per/%s" % (name, k)])))
        assert float(gt.abs().max()) > 0


def test_large_social_group_vs_reference_golden(setup):
    """SocialGroupLoss on a group of 28 / 30 members: above 25 torch.cdist takes its matmul form
    (guidance_loss.py:786); tests/golden/social_group_large.npz, written by oracle/gen_golden_social_large.py."""
    from oracle.gen_golden_social_large import APS, large_configs
    cfg = setup[0]
    batch = syn.make_batch(cfg, APS, seed=3, extent=(0.9, 0.7, 1.75), dense=True)
    gold = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "social_group_large.npz")))
    traj0 = torch.tensor(gold["traj"])
    for name, cfgs in large_configs().items():
        guid = orc.Guidance(cfgs)
        for si, cur in enumerate(guid.scenes):
            for gi, (gname, _, _, fn) in enumerate(cur):
                key = "%s_scene_%03d_%02d" % (gname, si, gi)
                fn.draw = (torch.tensor(gold["%s/randint/%s" % (name, key)]).long(),
                           torch.tensor(gold["%s/rand/%s" % (name, key)]))
        tr = traj0.clone().requires_grad_()
        tot, per = guid.loss(tr, batch)
        gt, = torch.autograd.grad(tot, tr)
        assert torch.equal(tot.detach(), torch.tensor(gold[name + "/total"])), name
        assert torch.equal(gt, torch.tensor(gold[name + "/grad"])), name
        for k, v in per.items():
            assert torch.equal(torch.nan_to_num(v), torch.nan_to_num(torch.tensor(gold["%s/per/%s" % (name, k)])))
        assert float(gt.abs().max()) > 0
