"""CPU oracle for the TRACE guided sampler -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

A from-scratch PyTorch-CPU restatement of the reference's sampling algorithm
(functional, state-dict driven, no nn.Module).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported CPU baseline; ``trace_amd`` never imports it
and fails loudly when its HIP library is missing.

Pinned (tests/test_oracle_golden.py) against golden vectors emitted from the
reference itself by ``oracle/gen_golden.py`` (the reference holds no tests or
fixtures for this path).

Arithmetic lives in ATen (torch 2.10.0 in this image; the reference pins 1.10.2,
README.md:26): conv1d / group_norm / linear / matmul / cdist are called the same way
the reference calls them, so elementwise chains round identically.  The guidance
gradient is taken with autograd here (the HIP path uses hand-written adjoints; this
is what checks them).

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- #
# schedule  (tbsim/models/trace_helpers.py:330-341, tbsim/models/trace.py:170-203)
# ----------------------------------------------------------------------------- #
def schedule_tables(n_steps, s=0.008):
    steps = n_steps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas64 = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    betas = torch.tensor(betas64, dtype=torch.float32)
    alphas = 1.0 - betas
    acp = torch.cumprod(alphas, dim=0)
    acp_prev = torch.cat([torch.ones(1), acp[:-1]])
    post_var = betas * (1.0 - acp_prev) / (1.0 - acp)
    tab = OrderedDict()
    tab["betas"] = betas
    tab["alphas_cumprod"] = acp
    tab["sqrt_recip_alphas_cumprod"] = torch.sqrt(1.0 / acp)
    tab["sqrt_recipm1_alphas_cumprod"] = torch.sqrt(1.0 / acp - 1)
    tab["posterior_variance"] = post_var
    tab["posterior_log_variance_clipped"] = torch.log(torch.clamp(post_var, min=1e-20))
    # the reference mixes torch tensors with np.sqrt of a torch tensor here (trace.py:200-203)
    # (numpy's float32 sqrt differs from torch.sqrt by 1 ulp on some entries, so use numpy)
    tab["posterior_mean_coef1"] = betas * torch.from_numpy(np.sqrt(acp_prev.numpy())) / (1.0 - acp)
    tab["posterior_mean_coef2"] = (1.0 - acp_prev) * torch.from_numpy(np.sqrt(alphas.numpy())) / (1.0 - acp)
    tab["sqrt_alphas_over_one_minus_alphas_cumprod"] = torch.sqrt(acp / (1.0 - acp))
    tab["sqrt_recip_one_minus_alphas_cumprod"] = 1.0 / torch.sqrt(1.0 - acp)
    return tab


# ----------------------------------------------------------------------------- #
# denoiser  (tbsim/models/temporal.py:17-142, tbsim/models/trace_helpers.py:17-64)
# ----------------------------------------------------------------------------- #
def _sinusoidal(t, dim):
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -e)
    a = t[:, None] * freqs[None, :]
    return torch.cat((a.sin(), a.cos()), dim=-1)


def _conv_block(sd, p, x, n_groups):
    w = sd[p + ".block.0.weight"]
    y = F.conv1d(x, w, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    y = F.group_norm(y.unsqueeze(2), n_groups, sd[p + ".block.2.weight"], sd[p + ".block.2.bias"])
    return F.mish(y.squeeze(2))


def _res_block(sd, p, x, emb, n_groups):
    tb = F.linear(F.mish(emb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    h = _conv_block(sd, p + ".blocks.0", x, n_groups) + tb[:, :, None]
    h = _conv_block(sd, p + ".blocks.1", h, n_groups)
    if (p + ".residual_conv.weight") in sd:
        r = F.conv1d(x, sd[p + ".residual_conv.weight"], sd[p + ".residual_conv.bias"])
    else:
        r = x
    return h + r


def unet_forward(sd, cfg, x, cond, t):
    """x (M, T, C_in) scaled trajectory+map feats, cond (M, cond_dim), t (M,) long
    -> (M, T, 2).  tbsim/models/temporal.py:117-142."""
    g = cfg.n_groups
    h = x.transpose(1, 2)
    te = _sinusoidal(t, cfg.time_dim)
    te = F.linear(te, sd["model.time_mlp.1.weight"], sd["model.time_mlp.1.bias"])
    te = F.linear(F.mish(te), sd["model.time_mlp.3.weight"], sd["model.time_mlp.3.bias"])
    emb = torch.cat([te, cond], dim=-1)
    n_res = len(cfg.unet_dims) - 1
    skips = []
    for i in range(n_res):
        h = _res_block(sd, "model.downs.%d.0" % i, h, emb, g)
        h = _res_block(sd, "model.downs.%d.1" % i, h, emb, g)
        skips.append(h)
        if i < n_res - 1:
            h = F.conv1d(h, sd["model.downs.%d.2.conv.weight" % i],
                         sd["model.downs.%d.2.conv.bias" % i], stride=2, padding=1)
    h = _res_block(sd, "model.mid_block1", h, emb, g)
    h = _res_block(sd, "model.mid_block2", h, emb, g)
    for i in range(n_res - 1):
        h = torch.cat((h, skips.pop()), dim=1)
        h = _res_block(sd, "model.ups.%d.0" % i, h, emb, g)
        h = _res_block(sd, "model.ups.%d.1" % i, h, emb, g)
        h = F.conv_transpose1d(h, sd["model.ups.%d.2.conv.weight" % i],
                               sd["model.ups.%d.2.conv.bias" % i], stride=2, padding=1)
    h = _conv_block(sd, "model.final_conv.0", h, g)
    h = F.conv1d(h, sd["model.final_conv.1.weight"], sd["model.final_conv.1.bias"])
    return h.transpose(1, 2)


# ----------------------------------------------------------------------------- #
# conditioning  (tbsim/models/trace.py:243-339, trace_helpers.py:66-174,
#                tbsim/models/base_models.py:14-104, tbsim/utils/batch_utils.py:29-35)
# ----------------------------------------------------------------------------- #
def _mlp_ln_relu(sd, p, x):
    idx = 0
    while ("%s.%d.weight" % (p, idx + 1)) in sd and sd["%s.%d.weight" % (p, idx + 1)].dim() == 1:
        x = F.linear(x, sd["%s.%d.weight" % (p, idx)], sd["%s.%d.bias" % (p, idx)])
        x = F.layer_norm(x, x.shape[-1:], sd["%s.%d.weight" % (p, idx + 1)],
                         sd["%s.%d.bias" % (p, idx + 1)])
        x = F.relu(x)
        idx += 3
    return F.linear(x, sd["%s.%d.weight" % (p, idx)], sd["%s.%d.bias" % (p, idx)])


def _hist_features(pos, yaw, speed, extent, avail, add, div):
    Bn, Th, _ = pos.shape
    add = torch.tensor(add, dtype=torch.float32)
    div = torch.tensor(div, dtype=torch.float32)
    hvec = torch.cat([torch.cos(yaw), torch.sin(yaw)], dim=-1)
    lw = extent[:, :2].unsqueeze(1).expand(Bn, Th, 2)
    pos = (pos + add[:2][None, None]) / div[:2][None, None]
    speed = (speed.unsqueeze(-1) + add[2]) / div[2]
    lw = (lw + add[3:][None, None]) / div[3:][None, None]
    feat = torch.cat([pos, hvec, speed, lw, avail.unsqueeze(-1)], dim=-1)
    feat = torch.where(avail.unsqueeze(-1), feat, torch.zeros_like(feat))
    return feat.reshape(Bn, -1)


def _agent_hist(sd, cfg, b, avail):
    f = _hist_features(b["history_positions"], b["history_yaws"], b["history_speeds"],
                       b["extent"], avail, cfg.agent_hist_norm_add, cfg.agent_hist_norm_div)
    return _mlp_ln_relu(sd, "agent_hist_encoder.traj_mlp._model", f)


def _neighbor_hist(sd, cfg, b, avail):
    pos = b["all_other_agents_history_positions"]
    B, Nn, Th, _ = pos.shape
    f = _hist_features(pos.reshape(B * Nn, Th, 2),
                       b["all_other_agents_history_yaws"].reshape(B * Nn, Th, 1),
                       b["all_other_agents_history_speeds"].reshape(B * Nn, Th),
                       b["all_other_agents_extents"].reshape(B * Nn, 3),
                       avail.reshape(B * Nn, Th),
                       cfg.neighbor_hist_norm_add, cfg.neighbor_hist_norm_div)
    enc = _mlp_ln_relu(sd, "neighbor_hist_encoder.agt_hist_encoder.traj_mlp._model", f)
    enc = enc.reshape(B, Nn, -1)
    padded = avail.sum(dim=-1) == 0
    enc = torch.where(padded.unsqueeze(-1), torch.full_like(enc, -float("inf")), enc)
    pooled = enc.amax(dim=1)
    alone = padded.sum(dim=-1) == Nn
    return torch.where(alone.unsqueeze(-1), torch.zeros_like(pooled), pooled)


def get_aux_info(sd, cfg, batch, use_cfg):
    """``batch['map_grid_feat']`` (and ``'map_grid_feat_non_cond'`` when use_cfg) stand in
    for the ResNet-18 map encoder output, which cannot be built offline."""
    B = batch["history_positions"].shape[0]
    curr = torch.zeros(B, 4)
    curr[:, 2] = batch["curr_speed"]
    cond_in = torch.cat([_agent_hist(sd, cfg, batch, batch["history_availabilities"]),
                         _neighbor_hist(sd, cfg, batch,
                                        batch["all_other_agents_history_availabilities"])], dim=-1)
    aux = {"cond_feat": _mlp_ln_relu(sd, "process_cond_mlp._model", cond_in),
           "curr_states": curr,
           "map_grid_feat": batch["map_grid_feat"],
           "raster_from_agent": batch["raster_from_agent"]}
    if use_cfg:
        no_h = torch.zeros_like(batch["history_availabilities"], dtype=torch.bool)
        no_n = torch.zeros_like(batch["all_other_agents_history_availabilities"], dtype=torch.bool)
        nc_in = torch.cat([_agent_hist(sd, cfg, batch, no_h),
                           _neighbor_hist(sd, cfg, batch, no_n)], dim=-1)
        aux["non_cond_feat"] = _mlp_ln_relu(sd, "process_cond_mlp._model", nc_in)
        aux["map_grid_feat_non_cond"] = batch["map_grid_feat_non_cond"]
    return aux


# ----------------------------------------------------------------------------- #
# scaling, map query, dynamics
# ----------------------------------------------------------------------------- #
def _coef(cfg, inds):
    add = torch.tensor(np.asarray(cfg.norm_add, dtype=np.float32)[inds])[None, None]
    div = torch.tensor(np.asarray(cfg.norm_div, dtype=np.float32)[inds])[None, None]
    return add, div


def scale_traj(cfg, x, inds=(0, 1, 2, 3, 4, 5)):      # trace.py:415-430
    add, div = _coef(cfg, list(inds))
    return (x + add) / div


def descale_traj(cfg, x, inds=(0, 1, 2, 3, 4, 5)):    # trace.py:432-447
    add, div = _coef(cfg, list(inds))
    return x * div - add


def transform_points(points, mat):                    # geometry_utils.py:105-171
    mt = mat.transpose(-1, -2)
    return points @ mt[:, :2, :2] + mt[:, -1:, :2]


def query_feature_grid(pos, grid):                    # trace_helpers.py:270-319
    B, T, _ = pos.shape
    x = pos[..., 0].reshape(-1)
    y = pos[..., 1].reshape(-1)
    x = torch.clamp(x, 0, grid.shape[-1] - 1 - 1e-3)
    y = torch.clamp(y, 0, grid.shape[-2] - 1 - 1e-3)
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    g = grid.permute(0, 2, 3, 1)
    bi = torch.arange(B)[:, None].expand(B, T).reshape(-1)
    fa, fb, fc, fd = g[bi, y0, x0], g[bi, y1, x0], g[bi, y0, x1], g[bi, y1, x1]
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    nrm = 1.0 / ((x1f - x0f) * (y1f - y0f))
    wa = (x1f - x) * (y1f - y) * nrm
    wb = (x1f - x) * (y - y0f) * nrm
    wc = (x - x0f) * (y1f - y) * nrm
    wd = (x - x0f) * (y - y0f) * nrm
    out = wa.unsqueeze(1) * fa + wb.unsqueeze(1) * fb + wc.unsqueeze(1) * fc + wd.unsqueeze(1) * fd
    return out.reshape(B, T, -1)


def query_map_feats(cfg, x, grid, raster_from_agent):  # trace.py:341-368
    pos = descale_traj(cfg, x.detach())[:, :, :2]
    rp = transform_points(pos, raster_from_agent)
    _, H, W = cfg.image_shape
    rp = torch.stack([rp[..., 0] * (grid.shape[-1] / W), rp[..., 1] * (grid.shape[-2] / H)], dim=-1)
    return query_feature_grid(rp, grid)


def unicycle_rollout(cfg, init, actions):
    """mode='parallel' of tbsim/models/trace_helpers.py:387-475 (prefix sums written as
    products with a lower-triangular ones matrix, as the reference does)."""
    dt = cfg.step_time
    T = actions.shape[-2]
    tri = torch.tril(torch.ones(T + 1, T + 1))
    avg = torch.zeros(T, T + 1)
    ar = torch.arange(T)
    avg[ar, ar] = 0.5
    avg[ar, ar + 1] = 0.5
    acc = torch.clip(actions[..., :1], cfg.acce_bound[0], cfg.acce_bound[1])
    yawvel = actions[..., 1:]
    v_raw = tri @ torch.cat((init[:, 2:3].unsqueeze(1), acc * dt), dim=1)
    v_all = torch.clip(v_raw, cfg.vbound[0], cfg.vbound[1])
    v_avg = avg @ v_all
    with torch.no_grad():
        v_prev = v_all[:, :-1]
        bound = torch.minimum(cfg.max_steer * v_prev.abs(),
                              cfg.max_yawvel / torch.clip(v_prev.abs(), min=0.1))
        bound = torch.clip(bound, min=0.1)
    yv = torch.clip(yawvel, -bound, bound)
    yaw_all = tri @ torch.cat((init[:, 3:4].unsqueeze(1), yv * dt), dim=1)
    yaw_prev = yaw_all[:, :-1]
    vxy = torch.cat((v_avg * torch.cos(yaw_prev), v_avg * torch.sin(yaw_prev)), dim=-1)
    xy = (tri @ torch.cat((init[:, :2].unsqueeze(1), vxy * dt), dim=1))[:, 1:]
    return torch.cat((xy, v_all[:, 1:], yaw_all[:, 1:]), dim=-1)


def action_to_state_and_action(cfg, a, curr_states, scaled_input=True, descaled_output=False):
    """trace.py:390-413."""
    if scaled_input:
        a = descale_traj(cfg, a, (4, 5))
    s = unicycle_rollout(cfg, curr_states, a)
    out = torch.cat([s, a], dim=-1)
    if scaled_input and not descaled_output:
        out = scale_traj(cfg, out)
    return out


# ----------------------------------------------------------------------------- #
# guidance objectives  (tbsim/utils/guidance_loss.py)
# ----------------------------------------------------------------------------- #
def agents_to_world(pos, yaw, world_from_agent):      # trace_helpers.py:343-364
    B, N, T, _ = pos.shape
    M = world_from_agent.unsqueeze(1).expand(B, N, 3, 3).reshape(B * N, 3, 3)
    wp = transform_points(pos.reshape(B * N, T, 2), M).reshape(B, N, T, 2)
    h = torch.cat([torch.cos(yaw), torch.sin(yaw)], dim=-1).reshape(B * N, T, 2)
    wh = transform_points(h, M).reshape(B, N, T, 2) - M[:, :2, 2].reshape(B, N, 1, 2)
    return wp, torch.atan2(wh[..., 1], wh[..., 0]).unsqueeze(-1)


def _rot_rows(yaw):
    s, c = torch.sin(yaw).unsqueeze(-1), torch.cos(yaw).unsqueeze(-1)
    return torch.cat((torch.cat((c, s), dim=-1), torch.cat((-s, c), dim=-1)), dim=-2)


class TargetPos:                                       # guidance_loss.py:503-540
    def __init__(self, target_pos, min_target_time=0.0):
        self.target_pos = torch.as_tensor(target_pos, dtype=torch.float32)
        self.min_target_time = min_target_time

    def __call__(self, x, batch, mask):
        if mask is not None:
            x = x[mask]
        assert x.shape[0] == self.target_pos.shape[0]
        t0 = int(self.min_target_time * x.shape[2])
        d = x[:, :, t0:, :2] - self.target_pos[:, None, None]
        dist = torch.norm(d, dim=-1)
        w = F.softmin(dist, dim=-1)
        return torch.mean(w * torch.sum(d ** 2, dim=-1), dim=-1)


class TargetPosAtTime:                                 # guidance_loss.py:466-501
    def __init__(self, target_pos, target_time):
        self.target_pos = torch.as_tensor(target_pos, dtype=torch.float32)
        self.target_time = torch.as_tensor(target_time).long()

    def __call__(self, x, batch, mask):
        if mask is not None:
            x = x[mask]
        assert x.shape[0] == self.target_pos.shape[0] == self.target_time.shape[0]
        p = x[torch.arange(x.shape[0]), :, self.target_time, :2]
        return torch.norm(p - self.target_pos[:, None], dim=-1)


class TargetSpeed:                                     # guidance_loss.py:148-178
    def __init__(self, target_speed, dt, mode="use_action"):
        self.target_speed, self.dt, self.mode = target_speed, dt, mode

    def __call__(self, x, batch, mask):
        if mask is not None:
            x = x[mask]
        tgt = self.target_speed
        if isinstance(tgt, list):
            assert x.shape[0] == len(tgt)
            tgt = torch.tensor(tgt)
        if self.mode == "use_action":
            return torch.mean((x[..., 2] - tgt) ** 2, dim=-1)
        v = torch.norm((x[:, :, 1:, :2] - x[:, :, :-1, :2]) / self.dt, dim=-1)
        return torch.mean((v - tgt) ** 2, dim=-1)


class MinSpeed:                                        # guidance_loss.py:180-196
    def __init__(self, min_speed, dt):
        self.min_speed, self.dt = min_speed, dt

    def __call__(self, x, batch, mask):
        if mask is not None:
            x = x[mask]
        v = torch.norm((x[:, :, 1:, :2] - x[:, :, :-1, :2]) / self.dt, dim=-1)
        return torch.mean(F.gelu(self.min_speed - v) ** 2, dim=-1)


class AgentCollision:                                  # guidance_loss.py:202-338
    def __init__(self, num_disks=5, buffer_dist=0.2):
        self.num_disks, self.buffer_dist = num_disks, buffer_dist

    @staticmethod
    def disks(num_disks, extent):                      # :241-250
        rad = extent[:, 1] / 2.0
        lo = -(extent[:, 0] / 2.0) + rad
        hi = (extent[:, 0] / 2.0) - rad
        cx = torch.stack([torch.linspace(lo[i].item(), hi[i].item(), num_disks)
                          for i in range(extent.shape[0])], dim=0)
        return torch.stack([cx, torch.zeros_like(cx)], dim=2), rad

    @staticmethod
    def same_scene_other_agent(scene_index):           # :252-262
        _, sid = torch.unique_consecutive(scene_index, return_inverse=True)
        same = sid[:, None] == sid[None, :]
        return same & ~torch.eye(sid.shape[0], dtype=torch.bool)

    def __call__(self, x, batch, mask):
        wp, wy = agents_to_world(x[..., :2], x[..., 3:4], batch["world_from_agent"])
        if mask is not None:
            m = mask[:, None, None, None]
            wp = torch.where(m, wp, wp.detach())
            wy = torch.where(m, wy, wy.detach())
        B, N, T, _ = wp.shape
        D = self.num_disks
        cents, rad = self.disks(D, batch["extent"])
        pen_d = rad.view(B, 1) + rad.view(1, B) + self.buffer_dist
        c = cents[:, None, None].expand(B, N, T, D, 2) @ _rot_rows(wy) + wp.unsqueeze(-2)
        c = c.transpose(0, 2).reshape(T * N, B, D, 2)
        c1 = c.view(T * N, B, 1, D, 2).expand(T * N, B, B, D, 2).reshape(-1, D, 2)
        c2 = c.view(T * N, 1, B, D, 2).expand(T * N, B, B, D, 2).reshape(-1, D, 2)
        dist = torch.cdist(c1, c2).view(-1, D * D).min(dim=1)[0].view(T * N, B, B)
        hit = (dist <= pen_d[None]) & self.same_scene_other_agent(batch["scene_index"])[None]
        pen = torch.where(hit, 1.0 - dist / pen_d[None], torch.zeros_like(dist))
        pen = pen.reshape(T, N, B, B).sum(0).sum(-1).transpose(0, 1)
        return pen[mask] if mask is not None else pen


class SocialGroup:                                     # guidance_loss.py:744-814
    """Members of a group keep `social_dist` to one assigned neighbour each.  The reference draws
    torch.randint / torch.rand inside forward (:803-804); `draw` = (randint (T*N, G, 1), rand (T*N, G))
    injects them, else they are drawn here in the same order."""

    def __init__(self, leader_idx=0, social_dist=1.5, cohesion=0.8):
        assert 0.0 <= cohesion <= 1.0
        self.leader_idx, self.social_dist, self.p = leader_idx, social_dist, cohesion
        self.draw = None

    def __call__(self, x, batch, mask):
        wfa = batch["world_from_agent"]
        pos, yaw = x[..., :2], x[..., 3:4]
        idx = torch.arange(pos.shape[0])
        if mask is not None:
            wfa, pos, yaw, idx = wfa[mask], pos[mask], yaw[mask], idx[mask]
        wp, _ = agents_to_world(pos, yaw, wfa)
        lead = (idx == self.leader_idx)[:, None, None, None].expand_as(wp)   # batch index vs leader_idx: as the reference
        wp = torch.where(lead, wp.detach(), wp)
        G, N, T, _ = wp.shape
        flat = wp.transpose(0, 2).reshape(T * N, G, 2)
        dist = torch.cdist(flat, flat)
        eye = torch.eye(G).unsqueeze(0).expand_as(dist)
        dist = torch.where(eye.bool(), float("inf") * eye, dist)
        nearest = torch.argmin(dist, dim=-1)
        choices = torch.arange(G)[None].expand(G, G).masked_select(~torch.eye(G, dtype=torch.bool)).view(G, G - 1)
        choices = choices.unsqueeze(0).expand(T * N, G, G - 1)
        if self.draw is not None:
            ri, u = self.draw
        else:
            ri = torch.randint(0, G - 1, (T * N, G, 1))
            u = torch.rand((T * N, G))
        rand_nb = torch.gather(choices, 2, ri)[:, :, 0]
        nb = torch.where(u < self.p, rand_nb, nearest)
        d = torch.gather(dist, 2, nb.unsqueeze(-1))[..., 0].reshape(T, N, G).transpose(0, 2)
        return torch.mean((d - self.social_dist) ** 2, dim=-1)


class MapCollision:                                    # guidance_loss.py:341-464
    def __init__(self, num_points_lw=(10, 10)):
        self.local = torch.cartesian_prod(torch.linspace(-0.5, 0.5, num_points_lw[0]),
                                          torch.linspace(-0.5, 0.5, num_points_lw[1]))

        # test hook: (B, N, T, P, K) int64, for every off-road sample point the ROWS (on-road points, -1 padded) to
        # pair it with instead of the arg-min of the matmul-form cdist (first entry negative: keep amin).  Several rows
        # = exactly tied distances: amin's backward shares the gradient evenly between them, as their mean does.  The reference's amin at
        # guidance_loss.py:444-458 is chaotic at the ulp level; with the choice held fixed, loss and gradient are
        # smooth functions of the trajectory, so a kernel can be compared on every step (tests/test_gpu_parity.py).
        self.forced_argmin = None
        # test hook: (B, N, T, P) bool of the last call -- the off-road sample points of its active (partly off-road)
        # steps, rows outside `mask` False
        self.last_off = None

    def __call__(self, x, batch, mask):
        dmap, ext, rfa = batch["drivable_map"], batch["extent"], batch["raster_from_agent"]
        forced = self.forced_argmin
        n_all = x.shape[0]
        if mask is not None:
            x, dmap, ext, rfa = x[mask], dmap[mask], ext[mask], rfa[mask]
            if forced is not None:
                forced = forced[mask]
        _, H, W = dmap.shape
        B, N, T, _ = x.shape
        flat = x.reshape(-1, 6)
        lw = ext[:, None, None].expand(B, N, T, 3).reshape(-1, 3)[:, :2]
        diag = torch.sqrt(torch.sum(lw * lw, dim=-1))
        rfa_f = rfa[:, None, None].expand(B, N, T, 3, 3).reshape(-1, 3, 3)
        pts = (self.local.unsqueeze(0) * lw.unsqueeze(-2)) @ _rot_rows(flat[:, 3:4]) \
            + flat[:, :2].unsqueeze(-2)
        pix = transform_points(pts, rfa_f).reshape(B, N, T, -1, 2).long().detach()
        px = torch.clamp(pix[..., 0], 0, W - 1)
        py = torch.clamp(pix[..., 1], 0, H - 1)
        P = pts.shape[1]
        idx = (py * W + px).reshape(B, -1)
        off = ~torch.gather(dmap.reshape(B, -1), 1, idx).bool()      # metrics.py:216-240
        off = off.reshape(-1, P)
        cnt = off.sum(dim=-1)
        partial = ~((cnt == 0) | (cnt == P))
        rec = (off & partial.unsqueeze(-1)).reshape(B, N, T, P)
        if mask is not None:
            full = torch.zeros(n_all, N, T, P, dtype=torch.bool)
            full[mask] = rec
            rec = full
        self.last_off = rec
        o_off, o_pts, o_diag = off[partial], pts[partial], diag[partial]
        d = torch.cdist(o_pts, o_pts.clone().detach())
        d = torch.where(o_off.unsqueeze(-1).expand(-1, -1, P), torch.full_like(d, float("inf")), d)
        dmin = torch.amin(d, dim=1)
        if forced is not None:
            K = forced.shape[-1]
            f = forced.reshape(-1, P, K)[partial]                                    # (n, P cols, K)
            picked = torch.gather(d, 1, f.clamp(min=0).permute(0, 2, 1))             # (n, K, P): d[n, f[n, j, k], j]
            wgt = (f >= 0).permute(0, 2, 1).to(d.dtype)
            picked = torch.where(wgt > 0, picked, torch.zeros_like(picked))
            mean = (picked * wgt).sum(1) / wgt.sum(1).clamp(min=1.0)
            dmin = torch.where(f[..., 0] >= 0, mean, dmin)
        l = torch.where(o_off, 1.0 - dmin / o_diag.unsqueeze(1), torch.zeros_like(dmin)).sum(-1)
        out = torch.zeros(pts.shape[0])
        out[partial] = l
        return out.reshape(B, N, T).sum(-1)


def progress_loss(pos, tgt, urgency, tgt_time, pref_speed, dt, min_progress_dist=0.5):
    """guidance_loss.py:542-590."""
    final = torch.norm(pos[:, :, -1] - tgt[:, None], dim=-1)
    if tgt_time is not None:
        goal = tgt_time[:, None] * dt * pref_speed * (1.0 - urgency[:, None])
        return F.relu(final - goal)
    horizon_dist = pos.shape[2] * dt * pref_speed
    goal = torch.maximum(urgency * horizon_dist, torch.tensor([min_progress_dist]))
    init = torch.norm(pos[:, :, 0] - tgt[:, None], dim=-1)
    return F.relu(goal[:, None] - (init - final))


class GlobalTargetPos:                                 # guidance_loss.py:668-742
    def __init__(self, target_pos, urgency, pref_speed=1.42, dt=0.1, min_progress_dist=0.5):
        self.target_pos = torch.tensor(target_pos, dtype=torch.float32)
        self.urgency = torch.tensor(urgency, dtype=torch.float32)
        self.pref_speed, self.dt, self.min_progress_dist = pref_speed, dt, min_progress_dist

    def __call__(self, x, batch, mask):
        afw = batch["agent_from_world"]
        if mask is not None:
            x, afw = x[mask], afw[mask]
        assert x.shape[0] == self.target_pos.shape[0] == self.urgency.shape[0]
        local = transform_points(self.target_pos[:, None], afw)[:, 0]
        exact = torch.norm(local, dim=-1) < x.shape[2] * self.dt * self.pref_speed
        loss = torch.zeros(x.shape[0], x.shape[1])
        if (~exact).any():
            loss[~exact] = progress_loss(x[..., :2][~exact], local[~exact], self.urgency[~exact],
                                         None, self.pref_speed, self.dt, self.min_progress_dist)
        if exact.any():
            loss[exact] = TargetPos(local[exact])(x[exact], None, None)
        return loss


class GlobalTargetPosAtTime:                           # guidance_loss.py:592-666
    def __init__(self, target_pos, target_time, urgency, pref_speed=1.42, dt=0.1):
        self.target_pos = torch.tensor(target_pos, dtype=torch.float32)
        self.target_time = torch.tensor(target_time)
        self.urgency = torch.tensor(urgency, dtype=torch.float32)
        self.pref_speed, self.dt = pref_speed, dt
        self.global_t = 0

    def __call__(self, x, batch, mask):
        afw = batch["agent_from_world"]
        if mask is not None:
            x, afw = x[mask], afw[mask]
        assert x.shape[0] == self.target_pos.shape[0] == self.target_time.shape[0]
        local = transform_points(self.target_pos[:, None], afw)[:, 0]
        lt = self.target_time - self.global_t
        exact = (lt < x.shape[2]) & (lt >= 0)
        prog = (~exact) & (lt >= 0)
        loss = torch.zeros(x.shape[0], x.shape[1])
        if not bool(exact.all()):
            loss[prog] = progress_loss(x[..., :2][prog], local[prog], self.urgency[prog], lt[prog],
                                       self.pref_speed, self.dt)
        if exact.any():
            loss[exact] = TargetPosAtTime(local[exact], lt[exact])(x[exact], None, None)
        return loss


GUIDANCE = {                                           # guidance_loss.py:853-864
    "target_speed": TargetSpeed, "agent_collision": AgentCollision, "map_collision": MapCollision,
    "target_pos_at_time": TargetPosAtTime, "target_pos": TargetPos,
    "global_target_pos_at_time": GlobalTargetPosAtTime, "global_target_pos": GlobalTargetPos,
    "min_speed": MinSpeed, "social_group": SocialGroup,
}


class Guidance:
    """guidance_loss.py:866-928 (DiffuserGuidance): per scene, a list of
    {name, weight, params, agents} dicts."""

    def __init__(self, config_list):
        assert len(config_list) > 0
        self.scenes = []
        for scene_cfgs in config_list:
            cur = []
            for c in scene_cfgs:
                assert set(c.keys()) == {"name", "weight", "params", "agents"}
                cur.append((c["name"], c["weight"], c["agents"], GUIDANCE[c["name"]](**c["params"])))
            self.scenes.append(cur)

    def update(self, global_t=None):
        if global_t is not None:
            for cur in self.scenes:
                for _, _, _, fn in cur:
                    if hasattr(fn, "global_t"):
                        fn.global_t = global_t

    def names(self):
        return [[c[0] for c in cur] for cur in self.scenes]

    def loss(self, x, batch):
        B, N = x.shape[:2]
        _, sid = torch.unique_consecutive(batch["scene_index"], return_inverse=True)
        total = 0.0
        per = OrderedDict()
        for si, cur in enumerate(self.scenes):
            for gi, (name, weight, agents, fn) in enumerate(cur):
                mask = sid == si
                if agents is not None:
                    sel = torch.nonzero(mask, as_tuple=True)[0][agents]
                    mask = torch.zeros_like(mask)
                    mask[sel] = True
                l = fn(x, batch, mask)
                full = torch.full((B, N), float("nan"))
                full[mask] = l.detach().clone()
                per["%s_scene_%03d_%02d" % (name, si, gi)] = full
                total = total + torch.mean(l) * weight
        return total, per


def choose_action_from_guidance(num_agents, guide_names, guide_losses):
    """guidance_loss.py:15-41.  guide_names: per scene list of loss names."""
    idx = torch.zeros(num_agents, dtype=torch.long)
    stacked = torch.stack(list(guide_losses.values()), dim=2)
    k = 0
    for names in guide_names:
        cur = stacked[..., k:k + len(names)]
        k += len(names)
        valid = ~torch.isnan(cur.sum(dim=[1, 2]))
        score = torch.nansum(cur[valid], dim=-1)
        if any(n in ("agent_collision", "social_group") for n in names):
            choice = torch.argmin(score.sum(dim=0))
        else:
            choice = torch.argmin(score, dim=-1)
        idx[valid] = choice
    return idx


# ----------------------------------------------------------------------------- #
# sampler  (tbsim/models/trace.py:450-500, 537-774)
# ----------------------------------------------------------------------------- #
class OracleSampler:
    def __init__(self, cfg, state_dict):
        self.cfg = cfg
        self.sd = {k: v.detach().float() for k, v in state_dict.items()}
        self.tab = schedule_tables(cfg.n_diffusion_steps)
        self.guidance = None

    def set_guidance(self, config_list):
        self.guidance = None
        if config_list is not None and any(len(c) > 0 for c in config_list):
            self.guidance = Guidance(config_list)

    def clear_guidance(self):
        self.guidance = None

    # -- pieces ------------------------------------------------------------- #
    def _tab(self, name, t):
        return self.tab[name].gather(-1, t).reshape(-1, 1, 1)

    def denoise(self, x, t, aux, w):
        """p_mean_variance, trace.py:561-602.  Returns (mean, log_var, x0, x_t)."""
        cfg, sd = self.cfg, self.sd
        feats = query_map_feats(cfg, x, aux["map_grid_feat"], aux["raster_from_agent"])
        pred = unet_forward(sd, cfg, torch.cat([x, feats], dim=-1), aux["cond_feat"], t)
        x_t = x[..., 4:].detach()
        if w != 0.0:
            feats_u = query_map_feats(cfg, x, aux["map_grid_feat_non_cond"], aux["raster_from_agent"])
            pred_u = unet_forward(sd, cfg, torch.cat([x, feats_u], dim=-1), aux["non_cond_feat"], t)
            a = self._tab("sqrt_recip_one_minus_alphas_cumprod", t)
            b = self._tab("sqrt_alphas_over_one_minus_alphas_cumprod", t)
            eps_c = a * x_t - b * pred
            eps_u = a * x_t - b * pred_u
            eps = (1 + w) * eps_c - w * eps_u
            pred = self._tab("sqrt_recip_alphas_cumprod", t) * x_t \
                - self._tab("sqrt_recipm1_alphas_cumprod", t) * eps
        mean = self.posterior_mean(pred, x_t, t)
        return mean, self._tab("posterior_log_variance_clipped", t), pred, x_t

    def posterior_mean(self, x0, x_t, t):               # trace.py:552-559
        return self._tab("posterior_mean_coef1", t) * x0 + self._tab("posterior_mean_coef2", t) * x_t

    def guidance_grad(self, x_guide, wrt, batch, aux, num_samp):
        """trace.py:604-633: loss on descaled rollout of x_guide, gradient w.r.t. ``wrt``."""
        with torch.enable_grad():
            full = action_to_state_and_action(self.cfg, x_guide, aux["curr_states"], True, True)
            total, per = self.guidance.loss(full.reshape(-1, num_samp, full.shape[1], 6), batch)
            grad, = torch.autograd.grad(total, wrt)
        return grad, per

    def p_sample(self, x, t, batch, aux, num_samp, w, noise, apply_guidance=True,
                 guide_clean=False, eval_final=False):
        """One reverse step, trace.py:635-706.  ``aux`` is already repeated per sample."""
        guided = self.guidance is not None and apply_guidance
        if guided and guide_clean:
            x = x.detach().requires_grad_()
            with torch.enable_grad():
                mean, logvar, x0, x_t = self.denoise(x, t, aux, w)
        else:
            with torch.no_grad():
                mean, logvar, x0, x_t = self.denoise(x, t, aux, w)
        nonzero = (1 - (t == 0).float()).reshape(-1, 1, 1)
        sigma = (0.5 * logvar).exp()
        grad = torch.zeros_like(mean)
        per = None
        if guided:
            if guide_clean:
                grad, per = self.guidance_grad(x0, x, batch, aux, num_samp)
                grad = grad[..., [4, 5]]
            else:
                xg = mean.clone().detach().requires_grad_()
                grad, per = self.guidance_grad(xg, xg, batch, aux, num_samp)
            grad = nonzero * grad
        noise = nonzero * sigma * noise
        with torch.no_grad():
            if self.guidance is not None and guide_clean:
                mean = self.posterior_mean(x0.detach() - grad, x_t, t)
                out = mean + noise
            else:
                out = mean.detach() - grad + noise
        if self.guidance is not None and eval_final:
            xe = out.clone().detach().requires_grad_()
            _, per = self.guidance_grad(xe, xe, batch, aux, num_samp)
        with torch.no_grad():
            out = action_to_state_and_action(self.cfg, out, aux["curr_states"])
        return out.detach(), per

    @staticmethod
    def repeat_aux(aux, n):
        return {k: v.repeat_interleave(n, dim=0) for k, v in aux.items()}

    def sample(self, batch, num_samp=1, class_free_guide_w=0.0, apply_guidance=True,
               guide_clean=False, noise_init=None, noise_steps=None, return_diffusion=True,
               n_steps=None):
        """DiffuserModel.forward, trace.py:450-500 + p_sample_loop :709-766.
        noise_init (B, N, T, 6), noise_steps (n_diffusion_steps, B*N, T, 2) -- entry i is
        consumed at diffusion step t = n-1-i.  n_steps truncates the loop (benchmark use)."""
        cfg = self.cfg
        B = batch["history_positions"].shape[0]
        T = cfg.horizon
        use_cfg = class_free_guide_w != 0.0
        with torch.no_grad():
            aux_b = get_aux_info(self.sd, cfg, batch, use_cfg)
        aux = self.repeat_aux(aux_b, num_samp)
        if noise_init is None:
            noise_init = torch.randn(B, num_samp, T, 6)
        x = noise_init.reshape(B * num_samp, T, 6)
        with torch.no_grad():
            x = action_to_state_and_action(cfg, x[..., [4, 5]], aux["curr_states"])
        diffusion = [x]
        steps = list(reversed(range(cfg.n_diffusion_steps)))
        if n_steps is not None:
            steps = steps[:n_steps]
        per = None
        for k, i in enumerate(steps):
            t = torch.full((B * num_samp,), i, dtype=torch.long)
            nz = noise_steps[k] if noise_steps is not None else torch.randn(B * num_samp, T, 2)
            x, per = self.p_sample(x, t, batch, aux, num_samp, class_free_guide_w, nz,
                                   apply_guidance, guide_clean, eval_final=(i == steps[-1]))
            if return_diffusion:
                diffusion.append(x)
        pred = x.reshape(B, num_samp, T, 6)
        traj = descale_traj(cfg, pred)
        out = {"pred_traj": pred, "trajectories": traj[..., [0, 1, 3]],
               "positions": traj[..., :2], "yaws": traj[..., 3:4],
               "curr_states": aux_b["curr_states"], "guide_losses": per}
        if return_diffusion:
            d = torch.stack([s.reshape(B, num_samp, T, 6) for s in diffusion], dim=3)
            out["diffusion_scaled"] = d
            out["diffusion_steps"] = descale_traj(cfg, d.reshape(B * num_samp, -1, 6)).reshape(d.shape)
        return out

    def get_action(self, batch, num_action_samples=1, class_free_guide_w=0.0,
                   guide_as_filter_only=False, guide_clean=False, step_index=0, **noise):
        """DiffuserTrafficModel.get_action, tbsim/algos/algos.py:261-309."""
        if self.guidance is not None:
            self.guidance.update(global_t=step_index)
        out = self.sample(batch, num_action_samples, class_free_guide_w,
                          apply_guidance=not guide_as_filter_only, guide_clean=guide_clean, **noise)
        B = out["positions"].shape[0]
        idx = torch.zeros(B, dtype=torch.long)
        if self.guidance is not None:
            idx = choose_action_from_guidance(B, self.guidance.names(), out["guide_losses"])
        ar = torch.arange(B)
        return {"positions": out["positions"][ar, idx], "yaws": out["yaws"][ar, idx],
                "act_idx": idx, "samples": out}
