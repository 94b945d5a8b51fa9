"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
vectors emitted by the reference.  Run on the MI355X box with ``-m gpu``.

Tolerances (BASELINE.json north_star): fp32, 1e-4 relative per teacher-forced step,
1e-3 m final ADE free-running.  "relative" is taken as |a-b| <= tol * (1 + |b|).

Carve-out (DESIGN.md "parity carve-out: map_collision"): the reference's MapCollisionLoss
gradient is chaotic at the 1-ulp level (torch.cdist's |a|^2+|b|^2-2ab form picks between tied
nearest on-road points by rounding noise): perturbing the reference's OWN input by ~1 ulp
changes ~25% of trajectory-steps by >1e-3.  For cases with map_collision the HIP path is
therefore compared where the reference is stable under such perturbations (stability mask
computed with the CPU oracle; the observed stable share and outlier count are asserted with a
margin), and the free-running ADE is only bounded loosely (< 1 m).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import trace_oracle as orc
from oracle.gen_golden import CASES, CHAOTIC, FULL_DIFFUSION, UNSTABLE, case_inputs, loss_component_configs
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float(((a - b).abs() / (1.0 + b.abs())).max())


def _to_dev(batch):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.fixture(scope="module")
def env():
    from trace_amd import _cabi
    from trace_amd.sampler import TraceSampler
    cfg = ModelConfig()
    sd = random_state_dict(cfg, 0)
    aps = [3, 2]
    batch = syn.make_batch(cfg, aps, seed=1, extent=(0.9, 0.7, 1.75), dense=True)
    sampler = TraceSampler(cfg, sd, DEV)
    oracle = orc.OracleSampler(cfg, sd)
    return dict(cfg=cfg, sd=sd, aps=aps, batch=batch, sampler=sampler, oracle=oracle,
                lib=_cabi.lib(), cabi=_cabi)


def test_native_library_loaded(env):
    assert env["lib"].trace_abi_version() == 1
    assert torch.cuda.is_available()


def test_dynamics_rollout_and_adjoint(env):
    cfg, s, L, cabi = env["cfg"], env["sampler"], env["lib"], env["cabi"]
    g = torch.Generator().manual_seed(21)
    B, N, T = 6, 3, cfg.horizon
    M = B * N
    for amp in (1.0, 8.0):          # amp 8 drives every clip
        a_s = torch.randn(M, T, 2, generator=g) * amp
        curr = torch.zeros(B, 4)
        curr[:, 2] = torch.tensor([1.5, 29.0, -9.5, 0.02, 3.0, 12.0])
        wgt = torch.randn(M, T, 6, generator=g)
        a_ref = a_s.clone().requires_grad_()
        full = orc.action_to_state_and_action(cfg, a_ref, curr.repeat_interleave(N, 0), True, True)
        gref, = torch.autograd.grad((full * wgt).sum(), a_ref)
        scaled = orc.scale_traj(cfg, full.detach())
        a_d, curr_d = a_s.to(DEV), curr.to(DEV)
        out = torch.empty(M, T, 6, device=DEV)
        dyn = C.byref(s._dyn)
        cabi.check(L.trace_dyn_rollout(a_d.data_ptr(), curr_d.data_ptr(), out.data_ptr(), M, N, T, 1, 0,
                                       dyn, None))
        assert _rel(out, full) < 1e-5
        cabi.check(L.trace_dyn_rollout(a_d.data_ptr(), curr_d.data_ptr(), out.data_ptr(), M, N, T, 1, 1,
                                       dyn, None))
        assert _rel(out, scaled) < 1e-5
        ga = torch.empty(M, T, 2, device=DEV)
        wd = wgt.to(DEV)
        cabi.check(L.trace_dyn_adjoint(a_d.data_ptr(), curr_d.data_ptr(), wd.data_ptr(), ga.data_ptr(), M,
                                       N, T, dyn, None))
        assert _rel(ga, gref) < 1e-4


def test_map_query(env):
    cfg, s, batch = env["cfg"], env["sampler"], env["batch"]
    g = torch.Generator().manual_seed(22)
    B, N = 5, 2
    x = torch.randn(B * N, cfg.horizon, 6, generator=g)
    x[0, :, 0] += 30.0      # far outside the raster: exercises the clamp
    aux = s.get_aux_info(batch, True)
    ref = orc.query_map_feats(cfg, x, batch["map_grid_feat"].repeat_interleave(N, 0),
                              batch["raster_from_agent"].repeat_interleave(N, 0))
    _, _, x_in, _ = s.denoise(x, 50, aux, N, w=0.0)
    feats = x_in[:, 6:38].transpose(1, 2)
    assert _rel(feats, ref) < 1e-5
    assert torch.equal(x_in[:, :6].transpose(1, 2).cpu(), x)
    assert float(x_in[:, 38:].abs().max()) == 0.0


@pytest.mark.parametrize("w", [0.0, 0.5])
def test_denoiser_forward_and_input_vjp(env, w):
    cfg, s, o, batch, L, cabi = (env[k] for k in ("cfg", "sampler", "oracle", "batch", "lib", "cabi"))
    g = torch.Generator().manual_seed(23)
    B, N = 5, 3
    M = B * N
    x = torch.randn(M, cfg.horizon, 6, generator=g)
    for t in (99, 37, 0):
        aux_o = o.repeat_aux(orc.get_aux_info(o.sd, cfg, batch, True), N)
        xr = x.clone().requires_grad_()
        tt = torch.full((M,), t, dtype=torch.long)
        with torch.enable_grad():
            _, _, x0_ref, _ = o.denoise(xr, tt, aux_o, w)
        wgt = torch.randn(M, cfg.horizon, 2, generator=g)
        gref, = torch.autograd.grad((x0_ref * wgt).sum(), xr)
        aux = s.get_aux_info(batch, True)
        x0, pred, x_in, (ws, ws_bytes) = s.denoise(x, t, aux, N, w=w, save=True)
        # the CFG detour x0 -> eps -> x0 multiplies fp32 rounding noise of the two denoiser
        # outputs by sqrt_recipm1_alphas_cumprod[99] ~ 2029 at t=99; x0 is
        # then damped by posterior_mean_coef1[99] = 0.0156 before it reaches the state, which
        # the teacher-forced step tests below check at 1e-4.
        tol = 2e-3 if (w != 0.0 and t == 99) else 1e-4
        assert _rel(x0, x0_ref) < tol, (t, _rel(x0, x0_ref))
        # input VJP through CFG combine + denoiser, channels [4,5]
        R = pred.shape[0]
        gp = torch.empty(R, cfg.horizon, 2, device=DEV)
        gx = torch.empty(R, cfg.horizon, 2, device=DEV)
        sc = s._coefs(t)
        wd = wgt.to(DEV)
        cabi.check(L.trace_cfg_combine_adjoint(wd.data_ptr(), gp.data_ptr(),
                                               gp[M:].data_ptr() if w else None, M, cfg.horizon, w,
                                               C.byref(sc), None))
        cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), R, gx.data_ptr(), ws.data_ptr(),
                                               ws_bytes, None))
        got = gx[:M] + gx[M:] if w else gx
        ref = gref[..., 4:6]
        scale = float(ref.abs().max())
        err = float((got.cpu() - ref).abs().max()) / scale
        assert err < 2e-4, (t, err)


def _fwd_vjp_errors(cfg, batch, sd, x, t, lib, cabi):
    """prediction / input-VJP error of both operand schemes against the CPU oracle (w = 0) on the rows of `x`."""
    from trace_amd.sampler import TraceSampler
    B, N = 5, 3
    M = B * N
    s, o = TraceSampler(cfg, sd, DEV), orc.OracleSampler(cfg, sd)
    g = torch.Generator().manual_seed(29)
    aux_o = o.repeat_aux(orc.get_aux_info(o.sd, cfg, batch, True), N)
    xr = x.clone().requires_grad_()
    tt = torch.full((M,), t, dtype=torch.long)
    with torch.enable_grad():
        _, _, x0_ref, _ = o.denoise(xr, tt, aux_o, 0.0)
    wgt = torch.randn(M, cfg.horizon, 2, generator=g)
    gref, = torch.autograd.grad((x0_ref * wgt).sum(), xr)
    ref = gref[..., 4:6]
    res = {}
    prev = lib.trace_conv_scheme(-1)
    try:
        for mode in (0, 1):
            lib.trace_conv_scheme(mode)
            aux = s.get_aux_info(batch, True)
            x0, pred, x_in, (ws, ws_bytes) = s.denoise(x, t, aux, N, w=0.0, save=True)
            gp = torch.empty(M, cfg.horizon, 2, device=DEV)
            gx = torch.empty(M, cfg.horizon, 2, device=DEV)
            sc = s._coefs(t)
            wd = wgt.to(DEV)
            cabi.check(lib.trace_cfg_combine_adjoint(wd.data_ptr(), gp.data_ptr(), None, M, cfg.horizon, 0.0, C.byref(sc), None))
            cabi.check(lib.trace_unet_backward_input(s._unet, gp.data_ptr(), M, gx.data_ptr(), ws.data_ptr(), ws_bytes, None))
            # per trajectory, against that trajectory's own scale
            e_x0 = ((x0.cpu() - x0_ref.detach()).abs().amax((1, 2)) / (1.0 + x0_ref.detach().abs().amax((1, 2))))
            e_g = ((gx.cpu() - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2)).clamp_min(1e-30))
            assert bool(torch.isfinite(x0).all()) and bool(torch.isfinite(gx).all())
            res[mode] = (e_x0, e_g)
    finally:
        lib.trace_conv_scheme(prev)
    return res


@pytest.mark.parametrize("widths", [(64, (2, 4, 8)), (32, (8, 8, 8))])
def test_denoiser_of_a_wider_unet(env, widths):
    """UNet widths 128 / 256 / 512 (base_dim = 64) and 256 / 256 / 256: a 128- / 256-channel head, GroupNorm groups of
    16 x 52 = 832 / 32 x 52 = 1664 elements, against the CPU oracle, forward and input-VJP, both operand schemes.
    Round 3 found the head's GroupNorm backward silently dropping everything past 448 elements of a group on such
    widths (input gradient off by its own size); the default width never exercised it.  (Widths above 512 are
    refused by trace_unet_create.)"""
    cfg = ModelConfig(base_dim=widths[0], dim_mults=widths[1])
    assert cfg.unet_dims[1] in (128, 256)
    sd = random_state_dict(cfg, 3)
    batch = syn.make_batch(cfg, env["aps"], seed=1, extent=(0.9, 0.7, 1.75), dense=True)
    x = torch.randn(15, cfg.horizon, 6, generator=torch.Generator().manual_seed(5))
    res = _fwd_vjp_errors(cfg, batch, sd, x, 37, env["lib"], env["cabi"])
    for mode, (e_x0, e_g) in res.items():
        assert float(e_x0.max()) < 1e-4, (mode, float(e_x0.max()))
        assert float(e_g.max()) < 2e-4, (mode, float(e_g.max()))


def test_fp16_pair_convs_on_heavy_tailed_weights(env):
    """Weights whose magnitudes spread over many octaves inside one layer, as a trained checkpoint's can (the weight
    scale 2^wexp is per LAYER, unet.hip:pack_conv, so a layer's small weights sit far below the scaled maximum of 2^14
    and their low fp16 terms go subnormal): 0.5 % of every conv weight tensor is multiplied by 2^8 .. 2^12.  Both operand
    schemes against the CPU oracle at the bar of test_denoiser_forward_and_input_vjp."""
    cfg, batch, L, cabi = (env[k] for k in ("cfg", "batch", "lib", "cabi"))
    g = torch.Generator().manual_seed(77)
    sd = dict(random_state_dict(cfg, 0))
    for k in list(sd):
        if k.startswith("model.") and sd[k].dim() == 3:
            w = sd[k].clone()
            hit = torch.rand(w.shape, generator=g) < 0.005
            w[hit] *= torch.pow(2.0, torch.randint(8, 13, (int(hit.sum()),), generator=g).float())
            sd[k] = w
    x = torch.randn(15, cfg.horizon, 6, generator=g)
    res = _fwd_vjp_errors(cfg, batch, sd, x, 37, L, cabi)
    for mode, (e_x0, e_g) in res.items():
        print("heavy-tailed weights, scheme %d: x0 %.2e, input-VJP %.2e (worst trajectory)" % (mode, float(e_x0.max()), float(e_g.max())))
    e_x0, e_g = res[1]
    # the un-normalised residual path carries 2^12-fold weights through twelve blocks: activations reach 1e18 here
    # (round 3: the operand-scale clamp at 2^40 turned them into fp16 infinities; it now spans the fp32 range)
    assert float(e_x0.max()) < 1e-4 and float(e_g.max()) < 2e-4, (float(e_x0.max()), float(e_g.max()))
    # a layer whose largest |weight| is beyond what the per-layer exponent covers (>= 2^24) has no fp16 stream and
    # runs on the bf16-triple kernel inside an otherwise fp16-pair pass
    sd2 = dict(random_state_dict(cfg, 0))
    sd2["model.mid_block1.blocks.0.block.0.weight"] = sd2["model.mid_block1.blocks.0.block.0.weight"] * 2.0 ** 27
    res2 = _fwd_vjp_errors(cfg, batch, sd2, x, 37, L, cabi)
    e_x0, e_g = res2[1]
    print("one layer x 2^27, scheme 1: x0 %.2e, input-VJP %.2e" % (float(e_x0.max()), float(e_g.max())))
    assert float(e_x0.max()) < 1e-4 and float(e_g.max()) < 2e-4, (float(e_x0.max()), float(e_g.max()))


@pytest.mark.parametrize("octaves", [18, 22])
def test_fp16_pair_convs_on_a_wide_activation_range_inside_one_group(env, octaves):
    """Forward activations whose magnitudes differ by more than 2^17 INSIDE one group of 16 trajectories (the activation
    scale of the fp16-pair scheme is per group, conv_bf3_kernels.h): every third trajectory's input is multiplied by
    2^octaves, which the first block's 1 x 1 residual conv carries past the GroupNorm into the block's output -- the
    next conv then sees trajectories 2^octaves apart under one scale, and the small ones lose low-term bits (their
    error stays at 2^-25 of the SCALED unit, i.e. grows relative to them).  Asserted: the large trajectories keep the
    usual bar; the small ones degrade gracefully (printed, bounded), and trace_hip.h states the range as a precondition."""
    cfg, batch, L, cabi = (env[k] for k in ("cfg", "batch", "lib", "cabi"))
    g = torch.Generator().manual_seed(78)
    x = torch.randn(15, cfg.horizon, 6, generator=g)
    big = torch.arange(15) % 3 == 0
    x[big] *= 2.0 ** octaves
    res = _fwd_vjp_errors(cfg, batch, dict(random_state_dict(cfg, 0)), x, 37, L, cabi)
    for mode, (e_x0, e_g) in res.items():
        print("2^%d spread, scheme %d: x0 big %.2e small %.2e; input-VJP big %.2e small %.2e" % (
            octaves, mode, float(e_x0[big].max()), float(e_x0[~big].max()), float(e_g[big].max()), float(e_g[~big].max())))
    e_x0, e_g = res[1]
    assert float(e_x0[big].max()) < 1e-4 and float(e_g[big].max()) < 2e-4
    assert float(e_x0[~big].max()) < 1e-3 and float(e_g[~big].max()) < 2e-3


# observed shares / errors of the map-collision stability mask, asserted with a margin (set after the first GPU run
# of round 3 printed them; round 2 had the loose 0.3 / 2e-3)
MAPC_STABLE_MIN = 0.75     # observed 0.878
MAPC_STABLE_ERR = 1e-4     # observed 5.8e-6 (oracle) / 1.9e-6 (golden)


def test_guidance_objectives_value_and_grad(env):
    cfg, aps, batch = env["cfg"], env["aps"], env["batch"]
    from trace_amd.guidance import DiffuserGuidance
    comp = dict(np.load(os.path.join(GOLDEN, "components.npz")))
    traj0 = torch.tensor(comp["loss/traj"])
    B, N, T = traj0.shape[:3]
    for lname, cfgs in loss_component_configs(batch, aps).items():
        tr = traj0.clone().requires_grad_()
        guid = orc.Guidance(cfgs)
        guid.update(global_t=7)
        tot, per = guid.loss(tr, batch)
        gref, = torch.autograd.grad(tot, tr)
        dg = DiffuserGuidance(cfgs)
        dg.update(global_t=7)
        plan = dg.build_plan(batch, N, T, DEV)
        td = traj0.reshape(B * N, T, 6).to(DEV).contiguous()
        gt = torch.zeros(B * N, T, 6, device=DEV)
        ll = plan.evaluate(td, gt, None)
        losses = plan.guide_losses(ll)
        for k, v in per.items():
            got = losses[k].cpu()
            assert torch.equal(torch.isnan(got), torch.isnan(v)), (lname, k)
            assert _rel(torch.nan_to_num(got), torch.nan_to_num(v)) < 1e-4, (lname, k)
        scale = max(float(gref.abs().max()), 1e-12)
        got = gt.cpu().reshape(B, N, T, 6)
        gold = torch.tensor(comp["loss/%s/grad" % lname])
        if lname == "map_collision":
            # (i) EVERY cell, with the arg-min choice taken out of the comparison: the kernel reports which on-road
            # point it paired each off-road point with (trace_guide_map_collision_argmin), the oracle's amin is
            # replaced by that choice (guidance_loss.py:431-458 with a forced arg-min), and loss + gradient must
            # then agree at 1e-4 on all (agent, sample, step) cells -- what differs between the two is only WHICH
            # of two nearly equidistant points won, never the arithmetic given the winner.
            plan2 = dg.build_plan(batch, N, T, DEV)
            plan2.debug_map_argmin = True
            gt2 = torch.zeros(B * N, T, 6, device=DEV)
            ll2 = plan2.evaluate(td, gt2, None)
            assert torch.equal(gt2, gt) and all(torch.equal(a, b) for a, b in zip(ll, ll2))
            (am,) = plan2.map_argmin.values()
            rows, nt = plan2.decode_map_argmin(am)
            forced = rows.reshape(B, N, T, -1, 4)
            ties = (nt.reshape(B, N, T, -1) > 4).any(-1)             # more tied rows than the diagnostic reports
            for scene in guid.scenes:
                for _, _, _, fn in scene:
                    fn.forced_argmin = forced
            trf = traj0.clone().requires_grad_()
            tot_f, per_f = guid.loss(trf, batch)
            gforced, = torch.autograd.grad(tot_f, trf)
            for scene in guid.scenes:
                for _, _, _, fn in scene:
                    fn.forced_argmin = None
            for k, v in per_f.items():
                assert _rel(torch.nan_to_num(losses[k].cpu()), torch.nan_to_num(v)) < 1e-4, (lname, k)
            active_f = gforced.abs().amax(-1) > 0
            assert int(active_f.sum()) >= 90 and int(ties.sum()) == 0, (int(active_f.sum()), int(ties.sum()))
            n_tied = int((nt > 1).sum())
            err_f = float(((got - gforced).abs().amax(-1) * (~ties)).max()) / scale
            # share of active cells where the kernel's choice differs from the oracle's own amin
            flipped = ((gforced - gref).abs().amax(-1) / scale > 1e-4) & active_f
            print("map_collision: %d active cells, %d off-road points with exactly tied nearest points, forced-arg-min "
                  "gradient error %.2e, kernel's choice differs from the oracle's amin on %d cells"
                  % (int(active_f.sum()), n_tied, err_f, int(flipped.sum())))
            assert err_f < 1e-4, err_f
            # (ii) as before: the cells (agent, sample, step) where the oracle agrees with itself under
            # ~1-ulp perturbations of the input trajectory
            pg = torch.Generator().manual_seed(5)
            stable = torch.ones(B, N, T, dtype=torch.bool)
            for _ in range(6):
                trp = (traj0 * (1 + (torch.rand(traj0.shape, generator=pg) - 0.5) * 6e-7)).requires_grad_()
                tot_p, _ = guid.loss(trp, batch)
                gp, = torch.autograd.grad(tot_p, trp)
                stable &= ((gp - gref).abs().amax(-1) / scale) < 1e-3
            active = gref.abs().amax(-1) > 0
            frac = float((stable & active).sum()) / max(float(active.sum()), 1.0)
            err = float(((got - gref).abs().amax(-1) * stable).max()) / scale
            err_g = float(((got - gold).abs().amax(-1) * stable).max()) / scale
            print("map_collision: stable share %.3f, error on the stable cells %.2e (oracle) %.2e (golden)" % (frac, err, err_g))
            assert frac > MAPC_STABLE_MIN, frac
            assert err < MAPC_STABLE_ERR and err_g < MAPC_STABLE_ERR, (lname, err, err_g, frac)
            continue
        err = float((got - gref).abs().max()) / scale
        err_g = float((got - gold).abs().max()) / scale      # golden (reference) values too
        assert err < 2e-4 and err_g < 2e-4, (lname, err, err_g)


def _run_case(env_unused, name, teacher_forced):
    from trace_amd.sampler import TraceSampler
    c = case_inputs(name)
    cfg = c["cfg"]
    s = TraceSampler(cfg, random_state_dict(cfg, c["sd_seed"]), DEV)
    s.set_guidance(c["guidance"], c["batch"])
    s.update_guidance(global_t=c["step_index"])
    return c, s


@pytest.mark.parametrize("name", sorted(FULL_DIFFUSION))
def test_teacher_forced_steps_vs_reference_golden(name):
    """Every reverse step started from the REFERENCE's x_t (golden diffusion chain)."""
    c, s = _run_case(None, name, True)
    cfg = c["cfg"]
    gold = dict(np.load(os.path.join(GOLDEN, "sample_%s.npz" % name)))
    d = torch.tensor(gold["diffusion_steps"])              # (B,N,T,101,6) descaled
    B, N, T = d.shape[:3]
    dsc = orc.scale_traj(cfg, d.reshape(B * N, -1, 6)).reshape(B * N, T, 101, 6)
    # Two kinds of cases cannot be compared on every trajectory-step:
    #   CHAOTIC  (map_collision): the reference's gradient flips under 1-ulp input changes;
    #   UNSTABLE (noisy guidance at the eval-config weights): a reverse step amplifies fp32-level differences
    #            of its input several hundred times (the oracle moves by up to 1.4e-3 under a 3e-6 perturbation).
    # For both, the comparison runs on the trajectory-steps where the reference algorithm (CPU oracle, bit-equal
    # to the golden on the unperturbed input) reproduces itself under such perturbations of x_t.
    masked = name in CHAOTIC or name in UNSTABLE
    eps = 6e-6 if name in UNSTABLE else 6e-7
    if masked:
        o = orc.OracleSampler(cfg, random_state_dict(cfg, c["sd_seed"]))
        o.set_guidance(c["guidance"])
        o.guidance.update(global_t=c["step_index"])
        aux = o.repeat_aux(orc.get_aux_info(o.sd, cfg, c["batch"], c["w"] != 0.0), N)
        pg = torch.Generator().manual_seed(9)
    worst, n_stable, n_out, n_tot, worst_any = 0.0, 0, 0, 0, 0.0
    for k in range(cfg.n_diffusion_steps):
        t = cfg.n_diffusion_steps - 1 - k
        out = s.forward(c["batch"], num_samp=N, class_free_guide_w=c["w"], apply_guidance=c["apply"],
                        guide_clean=c["clean"], x_init=dsc[:, :, k], steps=[t],
                        noise_steps=c["noise_steps"][k:k + 1])
        got = out["pred_traj_scaled"].reshape(B * N, T, 6).cpu()
        ref = dsc[:, :, k + 1]
        rel = ((got - ref).abs() / (1.0 + ref.abs())).amax(dim=(1, 2))
        worst_any = max(worst_any, float(rel.max()))
        if not masked:
            worst = max(worst, float(rel.max()))
            continue
        stable = torch.ones(B * N, dtype=torch.bool)
        tt = torch.full((B * N,), t, dtype=torch.long)
        for _ in range(4):
            xp = dsc[:, :, k] * (1 + (torch.rand(dsc[:, :, k].shape, generator=pg) - 0.5) * eps)
            op, _ = o.p_sample(xp, tt, c["batch"], aux, N, c["w"], c["noise_steps"][k], c["apply"],
                               c["clean"], eval_final=False)
            stable &= ((op - ref).abs() / (1.0 + ref.abs())).amax(dim=(1, 2)) < 5e-5
        n_tot += B * N
        n_stable += int(stable.sum())
        n_out += int((stable & (rel > 1e-4)).sum())
    if not masked:
        assert worst < 1e-4, worst
    else:
        print("%s: stable traj-steps %d/%d, HIP outliers on stable ones %d, worst over all traj-steps %.2e"
              % (name, n_stable, n_tot, n_out, worst_any))
        # observed (round 2): clean_all 404/1000 stable, 0 outliers; noisy_all 547/1000, 8 outliers;
        # noisy_tp_ac_cfgw 961/1000, 0 outliers (worst over ALL its trajectory-steps 5.4e-4)
        lo, frac_out = {"clean_all": (0.35, 0.01), "noisy_all": (0.48, 0.025),
                        "noisy_tp_ac_cfgw": (0.9, 0.005)}[name]
        assert n_stable >= lo * n_tot, (n_stable, n_tot)
        # a flip the 4 oracle probes all missed can still occur; bound their share
        assert n_out <= frac_out * n_stable, (n_out, n_stable)


@pytest.mark.parametrize("name", sorted(CHAOTIC))
def test_teacher_forced_steps_with_the_arg_min_held_fixed(name):
    """The map-collision cases on EVERY trajectory-step.  `MapCollisionLoss` (guidance_loss.py:431-458) pairs each
    off-road sample point with its nearest on-road point through a matmul-form cdist whose 1e-4 .. 1e-3 relative noise
    makes that arg-min flip under 1-ulp changes of the input -- which is why
    `test_teacher_forced_steps_vs_reference_golden` can only compare the steps on which the reference reproduces
    itself.  Here the choice is taken out of the comparison instead of masked: each reverse step runs on the GPU from
    the reference's x_t and reports the pairs the kernel chose (trace_guide_map_collision_argmin); the CPU oracle --
    bit-equal to the reference on the goldens -- runs the same step with its amin replaced by those pairs; the two
    results must agree at the 1e-4 bar of every other case on the trajectories of all 100 steps, with two exactly
    identified, counted and bounded exceptions (below): pixel-edge reclassification of a sample point (<= 1 % of the
    trajectory-steps, observed 0.1 %) and pairs whose matmul-form distance is ill-conditioned in the reference's own
    arithmetic (<= 1 % above 1e-4, none above 5e-4, observed 0.6 % of noisy_all)."""
    c, s = _run_case(None, name, True)
    cfg = c["cfg"]
    gold = dict(np.load(os.path.join(GOLDEN, "sample_%s.npz" % name)))
    d = torch.tensor(gold["diffusion_steps"])
    B, N, T = d.shape[:3]
    dsc = orc.scale_traj(cfg, d.reshape(B * N, -1, 6)).reshape(B * N, T, 101, 6)
    o = orc.OracleSampler(cfg, random_state_dict(cfg, c["sd_seed"]))
    o.set_guidance(c["guidance"])
    o.guidance.update(global_t=c["step_index"])
    aux = o.repeat_aux(orc.get_aux_info(o.sd, cfg, c["batch"], c["w"] != 0.0), N)
    mapc = [fn for scene in o.guidance.scenes for _, _, _, fn in scene if isinstance(fn, orc.MapCollision)]
    assert mapc
    s.debug_map_argmin = True
    worst, n_tie_traj, n_pairs, n_flipped_vs_gold, n_reclass, worst_reclass = 0.0, 0, 0, 0, 0, 0.0
    over = []
    try:
        for k in range(cfg.n_diffusion_steps):
            t = cfg.n_diffusion_steps - 1 - k
            out = s.forward(c["batch"], num_samp=N, class_free_guide_w=c["w"], apply_guidance=c["apply"],
                            guide_clean=c["clean"], x_init=dsc[:, :, k], steps=[t],
                            noise_steps=c["noise_steps"][k:k + 1])
            got = out["pred_traj_scaled"].reshape(B * N, T, 6).cpu()
            (am,) = out["guidance_plan"].map_argmin.values()
            rows, nt = out["guidance_plan"].decode_map_argmin(am)
            n_pairs += int((nt > 0).sum())
            tie_traj = (nt.reshape(B * N, -1) > 4).any(-1)              # more tied rows than the diagnostic reports
            forced = rows.reshape(B, N, T, -1, 4)
            for fn in mapc:
                fn.forced_argmin = forced if t != 0 else None            # t = 0: guidance is evaluated, not applied
            tt = torch.full((B * N,), t, dtype=torch.long)
            op, _ = o.p_sample(dsc[:, :, k], tt, c["batch"], aux, N, c["w"], c["noise_steps"][k], c["apply"],
                               c["clean"], eval_final=False)
            rel = ((got - op).abs() / (1.0 + op.abs())).amax(dim=(1, 2))
            # the loss has a second discontinuity that no arg-min fixes: WHICH sample points are off-road comes from a
            # truncated pixel index (guidance_loss.py:406-410), so a point within an ulp of a pixel edge can be
            # classified differently from the two sides' (1-ulp different) trajectories.  Such trajectory-steps are
            # identified exactly -- the kernel's off-road set differs from the oracle's -- counted, and bounded.
            off_o = torch.zeros_like(mapc[0].last_off)
            for fn in mapc:
                off_o |= fn.last_off
            reclass = ((nt.reshape(B, N, T, -1) > 0) != off_o).any(-1).any(-1).reshape(B * N)
            if t == 0:
                reclass[:] = False
            n_reclass += int(reclass.sum())
            n_tie_traj += int(tie_traj.sum())
            relc = rel * (~tie_traj) * (~reclass)
            worst = max(worst, float(relc.max()))
            if t != 0 and bool((relc > 1e-4).any()):
                # how far does the ORACLE itself move, pairs held fixed, when its input moves by an ulp?  (the
                # matmul-form distance |a|^2 + |b|^2 - 2ab of a pair 0.1 m apart cancels ~3 digits, and it sits in
                # the gradient's denominator: smooth, but ill-conditioned)
                pg = torch.Generator().manual_seed(1000 + k)
                xp = dsc[:, :, k] * (1 + (torch.rand(dsc[:, :, k].shape, generator=pg) - 0.5) * 6e-7)
                op2, _ = o.p_sample(xp, tt, c["batch"], aux, N, c["w"], c["noise_steps"][k], c["apply"], c["clean"],
                                    eval_final=False)
                sens = ((op2 - op).abs() / (1.0 + op.abs())).amax(dim=(1, 2))
                for m in torch.nonzero(relc > 1e-4)[:, 0].tolist():
                    over.append((t, m, float(relc[m]), float(sens[m])))
            worst_reclass = max(worst_reclass, float((rel * reclass).max()))
            ref = dsc[:, :, k + 1]
            n_flipped_vs_gold += int((((op - ref).abs() / (1.0 + ref.abs())).amax(dim=(1, 2)) > 1e-4).sum())
    finally:
        for fn in mapc:
            fn.forced_argmin = None
    print("%s: %d (off-road point, on-road point) pairs over 100 steps, worst trajectory-step with the kernel's pairs "
          "%.2e, trajectory-steps with more than 4 tied rows %d, trajectory-steps whose off-road point SET differs "
          "(pixel-edge reclassification) %d (worst %.2e), trajectory-steps on which the kernel's pairs differ from the "
          "reference's own (golden) choice %d / %d" % (name, n_pairs, worst, n_tie_traj, n_reclass, worst_reclass,
                                                       n_flipped_vs_gold, 100 * B * N))
    print("trajectory-steps above 1e-4 (t, trajectory, HIP vs oracle, oracle vs itself 1 ulp away):", over)
    assert n_pairs > 1000, n_pairs                      # the loss is active
    assert n_tie_traj == 0, n_tie_traj
    assert n_reclass <= 0.01 * 100 * B * N, n_reclass   # observed: 1 / 1000 (clean_all), 0 (noisy_all)
    # observed: clean_all 0 trajectory-steps above 1e-4 (worst 3.6e-5); noisy_all 6 / 1000 between 1.1e-4 and 2.7e-4,
    # on which the oracle with the SAME pairs moves by 2e-5 .. 3.5e-4 when its input moves by an ulp (printed above)
    assert len(over) <= 0.01 * 100 * B * N, over
    assert worst < 5e-4, worst


@pytest.mark.parametrize("name", CASES)
def test_free_running_sample_vs_reference_golden(name):
    c, s = _run_case(None, name, False)
    gold = dict(np.load(os.path.join(GOLDEN, "sample_%s.npz" % name)))
    N = c["num_samp"]
    out = s.forward(c["batch"], num_samp=N, return_diffusion=True, return_guidance_losses=True,
                    class_free_guide_w=c["w"], apply_guidance=c["apply"], guide_clean=c["clean"],
                    noise_init=c["noise_init"], noise_steps=c["noise_steps"])
    pos = out["predictions"]["positions"].cpu()
    assert bool(torch.isfinite(pos).all())
    # the returned reverse chain (written in place by the step kernels, sampler.py): its last entry IS the sample
    ds = out["predictions"]["diffusion_steps"].cpu()
    assert torch.equal(ds[:, :, :, -1, :2], pos)
    if name in UNSTABLE:
        # noisy guidance at the eval-config weights: the free-running chain is an unstable feedback loop in
        # the reference itself; its strict comparison is the teacher-forced test above
        return
    ade = float(torch.norm(pos - torch.tensor(gold["positions"]), dim=-1).mean())
    chaotic = name in CHAOTIC
    print("free-running ADE vs golden (%s): %.3g m" % (name, ade))
    if not chaotic:
        assert ade < 1e-3, ade
        if "diffusion_steps" in gold:         # every entry of the chain, not only the last
            gd = torch.tensor(gold["diffusion_steps"])
            assert gd.shape == ds.shape, (gd.shape, ds.shape)
            dd = float(torch.norm(ds[..., :2] - gd[..., :2], dim=-1).mean())
            print("free-running chain (all %d entries) vs golden (%s): %.3g m" % (ds.shape[3], name, dd))
            assert dd < 1e-3, dd
    else:
        # observed 0.12 m (clean_all) / 0.26 m (noisy_all); two reference runs 1 ulp apart end 0.28 m apart
        print("free-running ADE vs golden with map_collision guidance: %.4f m" % ade)
        assert ade < 1.0, ade
    if c["guidance"] is not None:
        gl = out["predictions"]["guide_losses"]
        for k in gl.keys():
            ref = torch.tensor(gold["guide_loss/" + k])
            got = gl[k].cpu()
            assert torch.equal(torch.isnan(got), torch.isnan(ref)), k
            if not chaotic:
                tol = 2e-2 if k.startswith("map_collision") else 5e-3
                assert _rel(torch.nan_to_num(got), torch.nan_to_num(ref)) < tol, (k, got, ref)
        idx = out["guidance_plan"].choose_action(out["guidance_layer_losses"]).cpu().numpy()
        if not chaotic:
            assert np.array_equal(idx, gold["act_idx"]), (idx, gold["act_idx"])


def test_social_group_value_and_grad(env):
    """SocialGroupLoss kernels against the reference's golden (loss per agent and gradient) with the
    reference's random neighbour draws injected; then the device-drawn path against the oracle fed the
    same picks."""
    from oracle.gen_golden_social import social_configs
    from trace_amd.guidance import DiffuserGuidance
    cfg, aps, batch = env["cfg"], env["aps"], env["batch"]
    gold = dict(np.load(os.path.join(GOLDEN, "social_group.npz")))
    traj0 = torch.tensor(dict(np.load(os.path.join(GOLDEN, "components.npz")))["loss/traj"])
    B, N, T = traj0.shape[:3]
    td = traj0.reshape(B * N, T, 6).to(DEV).contiguous()
    for name, cfgs in social_configs().items():
        plan = DiffuserGuidance(cfgs).build_plan(batch, N, T, DEV)
        plan.social_tape = {k[len(name) + len("/randint/"):]: (torch.tensor(gold[k]),
                                                              torch.tensor(gold[k.replace("/randint/", "/rand/")]))
                            for k in gold if k.startswith(name + "/randint/")}
        gt = torch.zeros(B * N, T, 6, device=DEV)
        losses = plan.guide_losses(plan.evaluate(td, gt, None))
        for k in plan.social_tape:
            want = torch.tensor(gold["%s/per/%s" % (name, k)])
            got = losses[k].cpu()
            assert torch.equal(torch.isnan(got), torch.isnan(want)), (name, k)
            assert _rel(torch.nan_to_num(got), torch.nan_to_num(want)) < 1e-5, (name, k)
        gref = torch.tensor(gold[name + "/grad"])
        scale = float(gref.abs().max())
        err = float((gt.cpu().reshape(B, N, T, 6) - gref).abs().max()) / scale
        assert scale > 0 and err < 2e-5, (name, err)
    # device-drawn neighbours: finite, and the leader never receives a gradient
    cfgs = social_configs()["mixed"]
    plan = DiffuserGuidance(cfgs).build_plan(batch, N, T, DEV)
    plan.generator = torch.Generator(device=DEV).manual_seed(3)
    gt = torch.zeros(B * N, T, 6, device=DEV)
    ll = plan.evaluate(td, gt, None)
    g = gt.reshape(B, N, T, 6)
    assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(ll[0]).all())
    assert float(g[0].abs().max()) == 0.0 and float(g[3].abs().max()) == 0.0      # leaders (batch idx 0, 3)
    assert float(g[1].abs().max()) > 0 and float(g[4].abs().max()) > 0
    assert float(g[..., 2:].abs().max()) == 0.0                                    # positions only


def test_large_social_group_value_and_grad(env):
    """A social group of 28 / 30 members (above 25 torch.cdist takes its matmul form, guidance_loss.py:786): the HIP
    kernels against the reference's golden with the reference's draws injected, next to a 2-member group in the
    exact form.  The matmul form carries ~1e-4 relative noise at these coordinates; the kernel restates ATen's
    operation order, so the nearest-neighbour choices -- and with them loss and gradient -- agree."""
    from oracle.gen_golden_social_large import APS, large_configs
    from trace_amd.guidance import DiffuserGuidance
    cfg = env["cfg"]
    batch = syn.make_batch(cfg, APS, seed=3, extent=(0.9, 0.7, 1.75), dense=True)
    gold = dict(np.load(os.path.join(GOLDEN, "social_group_large.npz")))
    traj0 = torch.tensor(gold["traj"])
    B, N, T = traj0.shape[:3]
    td = traj0.reshape(B * N, T, 6).to(DEV).contiguous()
    for name, cfgs in large_configs().items():
        plan = DiffuserGuidance(cfgs).build_plan(batch, N, T, DEV)
        widths = sorted(int(ly["dev"]["members"].shape[1]) for ly in plan.layers)
        assert widths == ([25, 32] if name == "large_mixed" else [32]), widths
        plan.social_tape = {k[len(name) + len("/randint/"):]: (torch.tensor(gold[k]).long(),
                                                              torch.tensor(gold[k.replace("/randint/", "/rand/")]))
                            for k in gold if k.startswith(name + "/randint/")}
        gt = torch.zeros(B * N, T, 6, device=DEV)
        losses = plan.guide_losses(plan.evaluate(td, gt, None))
        for k in plan.social_tape:
            want = torch.tensor(gold["%s/per/%s" % (name, k)])
            got = losses[k].cpu()
            assert torch.equal(torch.isnan(got), torch.isnan(want)), (name, k)
            assert _rel(torch.nan_to_num(got), torch.nan_to_num(want)) < 1e-5, (name, k, _rel(torch.nan_to_num(got), torch.nan_to_num(want)))
        gref = torch.tensor(gold[name + "/grad"])
        scale = float(gref.abs().max())
        err = float((gt.cpu().reshape(B, N, T, 6) - gref).abs().max()) / scale
        assert scale > 0 and err < 1e-4, (name, err)


def test_sampler_with_social_group_guidance(env):
    """Clean-guided sampling steps with social_group + target_pos guidance: HIP sampler vs CPU oracle
    with the same (static) neighbour draws injected on both sides."""
    from oracle.gen_golden_social import social_configs
    from trace_amd.sampler import TraceSampler
    cfg, aps, batch, sd = env["cfg"], env["aps"], env["batch"], env["sd"]
    gold = dict(np.load(os.path.join(GOLDEN, "social_group.npz")))
    guidance = syn.merge_guidance(social_configs()["mixed"], syn.target_pos_guidance(batch, aps, weight=300.0, seed=5))
    tape = {k[len("mixed/randint/"):]: (torch.tensor(gold[k]), torch.tensor(gold[k.replace("/randint/", "/rand/")]))
            for k in gold if k.startswith("mixed/randint/")}
    N, n_steps = 3, 10
    init, steps = syn.make_noise(cfg, sum(aps), N, seed=9)
    s = TraceSampler(cfg, sd, "cuda:0")
    s.set_guidance(guidance, batch)
    s.current_guidance.social_tape = tape
    out = s(batch, num_samp=N, class_free_guide_w=0.0, guide_clean=True, noise_init=init, noise_steps=steps,
            n_steps=n_steps, return_guidance_losses=True)
    torch.cuda.synchronize()
    o = orc.OracleSampler(cfg, sd)
    o.set_guidance(guidance)
    for si, cur in enumerate(o.guidance.scenes):
        for gi, (gname, _, _, fn) in enumerate(cur):
            if gname == "social_group":
                fn.draw = tape["%s_scene_%03d_%02d" % (gname, si, gi)]
    ref = o.sample(batch, N, 0.0, True, True, noise_init=init, noise_steps=steps, n_steps=n_steps,
                   return_diffusion=False)
    got = out["pred_traj_scaled"].cpu()
    err = float(((got - ref["pred_traj"]).abs() / (1 + ref["pred_traj"].abs())).max())
    assert err < 1e-4, err
    gl = out["predictions"]["guide_losses"]
    assert any(k.startswith("social_group") for k in gl.keys())


def test_split_agent_collision_configs_and_weight_zero_guides(env):
    """Two agent_collision configs with equal parameters and disjoint `agents` lists in ONE scene (the reference
    detaches the other config's agents, guidance_loss.py:272-282: cross-config pairs get the gradient once, not
    twice), and a guide of weight 0 (its loss is still evaluated, reported and used for selection)."""
    from trace_amd.guidance import DiffuserGuidance
    cfg, aps, batch = env["cfg"], env["aps"], env["batch"]
    comp = dict(np.load(os.path.join(GOLDEN, "components.npz")))
    traj0 = torch.tensor(comp["loss/traj"])
    B, N, T = traj0.shape[:3]
    ac = lambda w, agents: {"name": "agent_collision", "weight": w, "params": {"num_disks": 2, "buffer_dist": 0.3},
                            "agents": agents}
    cfgs = [[ac(1.5, [0, 1]), ac(2.0, [2])],
            [{"name": "target_pos", "weight": 0.0, "params": {"target_pos": [[3.0, 0.5], [2.0, -1.0]]}, "agents": None},
             ac(0.7, None)]]
    tr = traj0.clone().requires_grad_()
    guid = orc.Guidance(cfgs)
    tot, per = guid.loss(tr, batch)
    gref, = torch.autograd.grad(tot, tr)
    plan = DiffuserGuidance(cfgs).build_plan(batch, N, T, DEV)
    assert sum(ly["kind"] == "agent_collision" for ly in plan.layers) == 2       # scene 0's two configs kept apart
    td = traj0.reshape(B * N, T, 6).to(DEV).contiguous()
    gt = torch.zeros(B * N, T, 6, device=DEV)
    ll = plan.evaluate(td, gt, None)
    losses = plan.guide_losses(ll)
    assert float(torch.nan_to_num(per["target_pos_scene_001_00"]).abs().max()) > 0
    for k, v in per.items():
        got = losses[k].cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(v)), k
        assert _rel(torch.nan_to_num(got), torch.nan_to_num(v)) < 1e-4, k
    scale = float(gref.abs().max())
    err = float((gt.cpu().reshape(B, N, T, 6) - gref).abs().max()) / scale
    assert scale > 0 and err < 2e-4, err
    idx = plan.choose_action(ll).cpu()
    ref_idx = orc.choose_action_from_guidance(B, guid.names(), {k: v.detach() for k, v in per.items()})
    assert torch.equal(idx, ref_idx), (idx, ref_idx)
