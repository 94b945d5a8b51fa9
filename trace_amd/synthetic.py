"""Seeded synthetic agent-centric batches shaped like the reference's parsed trajdata
batch (input contract: SURVEY.md section 8(b); keys from
/root/reference/tbsim/utils/trajdata_utils.py:238-270).

No datasets or checkpoints are available offline, so tests, goldens and the benchmark
use ORCA-style scenes generated here: agents in a
30 m x 30 m square (orca_sim/gen_dataset.py:22), extent 0.8 x 0.8 x 1.75
(orca_dataset.py:226-232), speed U(1,2) m/s, constant-velocity 31-frame histories,
neighbours within 15 m, a 224 x 224 raster at 12 px/m with a few box obstacles,
``raster_from_agent`` = [[12,0,56],[0,12,112],[0,0,1]].

Everything is produced on the CPU with a ``torch.Generator`` so that the same seed
gives bitwise identical inputs here and on the GPU box.
"""
import math

import torch

from .config import ModelConfig


def _pose_mat(x, y, yaw):
    c, s = torch.cos(yaw), torch.sin(yaw)
    m = torch.zeros(x.shape[0], 3, 3)
    m[:, 0, 0], m[:, 0, 1], m[:, 0, 2] = c, -s, x
    m[:, 1, 0], m[:, 1, 1], m[:, 1, 2] = s, c, y
    m[:, 2, 2] = 1.0
    return m


def make_batch(cfg: ModelConfig, agents_per_scene, seed=0, extent=(0.8, 0.8, 1.75),
               area=30.0, neighbor_radius=15.0, max_neighbors=None, num_obstacles=3,
               dense=False, with_map=True, grid_seed=None):
    """agents_per_scene: list of ints.  Returns a dict of CPU tensors.

    dense=True packs agents on a jittered lattice ~1.2 m apart so agent-agent
    collision penalties are active (masks must be forced on)."""
    g = torch.Generator().manual_seed(seed)
    Th = cfg.hist_frames
    dt = cfg.step_time
    B = int(sum(agents_per_scene))
    scene_index = torch.cat([torch.full((a,), i, dtype=torch.long)
                             for i, a in enumerate(agents_per_scene)])
    pos_w = torch.zeros(B, 2)
    o = 0
    for a in agents_per_scene:
        if dense:
            side = int(math.ceil(math.sqrt(a)))
            ij = torch.stack(torch.meshgrid(torch.arange(side), torch.arange(side),
                                            indexing="ij"), -1).reshape(-1, 2)[:a].float()
            p = (ij - (side - 1) / 2.0) * 1.2 + (torch.rand(a, 2, generator=g) - 0.5) * 0.5
        else:
            p = (torch.rand(a, 2, generator=g) - 0.5) * area
        pos_w[o:o + a] = p
        o += a
    yaw_w = (torch.rand(B, generator=g) * 2 - 1) * math.pi
    speed = 1.0 + torch.rand(B, generator=g)
    ext = torch.tensor(extent, dtype=torch.float32)[None].repeat(B, 1)
    if ext[0, 0] != ext[0, 1]:
        # small per-agent variation for non-square extents
        ext[:, :2] *= 1.0 + 0.1 * (torch.rand(B, 2, generator=g) - 0.5)

    world_from_agent = _pose_mat(pos_w[:, 0], pos_w[:, 1], yaw_w)
    agent_from_world = torch.linalg.inv(world_from_agent)

    # world-frame constant-velocity histories, frame Th-1 is "now"
    k = torch.arange(Th, dtype=torch.float32) - (Th - 1)
    hv = torch.stack([torch.cos(yaw_w), torch.sin(yaw_w)], -1)
    hist_w = pos_w[:, None, :] + hv[:, None, :] * (speed[:, None, None] * dt * k[None, :, None])

    hist_pos = torch.zeros(B, Th, 2)
    hist_pos[:, :, 0] = speed[:, None] * dt * k[None, :]
    hist_yaw = torch.zeros(B, Th, 1)
    hist_speed = speed[:, None].repeat(1, Th)
    hist_avail = torch.ones(B, Th, dtype=torch.bool)
    # a few agents have a shorter observed history
    short = torch.rand(B, generator=g) < 0.25
    cut = torch.randint(1, Th - 1, (B,), generator=g)
    for b in range(B):
        if short[b]:
            hist_avail[b, :cut[b]] = False

    # neighbours
    d = torch.cdist(pos_w, pos_w)
    same = scene_index[:, None] == scene_index[None, :]
    nb = (d < neighbor_radius) & same & ~torch.eye(B, dtype=torch.bool)
    Nn = int(nb.sum(1).max().item()) if B > 1 else 0
    Nn = max(Nn, 1)
    if max_neighbors is not None:
        Nn = min(Nn, max_neighbors)
    n_pos = torch.zeros(B, Nn, Th, 2)
    n_yaw = torch.zeros(B, Nn, Th, 1)
    n_speed = torch.zeros(B, Nn, Th)
    n_avail = torch.zeros(B, Nn, Th, dtype=torch.bool)
    n_ext = torch.zeros(B, Nn, 3)
    for b in range(B):
        idx = torch.nonzero(nb[b], as_tuple=True)[0]
        if idx.numel() > Nn:
            idx = idx[torch.argsort(d[b, idx])[:Nn]]
        n = idx.numel()
        if n == 0:
            continue
        R = agent_from_world[b]
        pw = hist_w[idx]                                   # n, Th, 2
        n_pos[b, :n] = pw @ R[:2, :2].T + R[:2, 2]
        n_yaw[b, :n, :, 0] = (yaw_w[idx] - yaw_w[b])[:, None]
        n_speed[b, :n] = speed[idx][:, None]
        n_avail[b, :n] = hist_avail[idx]
        n_ext[b, :n] = ext[idx]

    C, H, W = cfg.image_shape
    rfa = torch.tensor([[12.0, 0.0, 56.0], [0.0, 12.0, 112.0], [0.0, 0.0, 1.0]])
    rfa = rfa * (H / 224.0)
    rfa[2, 2] = 1.0
    raster_from_agent = rfa[None].repeat(B, 1, 1)

    batch = {
        "history_positions": hist_pos, "history_yaws": hist_yaw, "history_speeds": hist_speed,
        "history_availabilities": hist_avail, "extent": ext, "curr_speed": speed.clone(),
        "all_other_agents_history_positions": n_pos, "all_other_agents_history_yaws": n_yaw,
        "all_other_agents_history_speeds": n_speed,
        "all_other_agents_history_availabilities": n_avail, "all_other_agents_extents": n_ext,
        "raster_from_agent": raster_from_agent, "world_from_agent": world_from_agent,
        "agent_from_world": agent_from_world, "scene_index": scene_index,
        "centroid": pos_w.clone(), "yaw": yaw_w.clone(),
    }

    if with_map:
        # per-scene axis-aligned world boxes, rasterised into each agent's frame
        drivable = torch.ones(B, H, W, dtype=torch.bool)
        py, px = torch.meshgrid(torch.arange(H, dtype=torch.float32),
                                torch.arange(W, dtype=torch.float32), indexing="ij")
        ax = (px - rfa[0, 2]) / rfa[0, 0]
        ay = (py - rfa[1, 2]) / rfa[1, 1]
        o = 0
        for si, a in enumerate(agents_per_scene):
            ctr = (torch.rand(num_obstacles, 2, generator=g) - 0.5) * area
            half = 1.5 + 3.5 * torch.rand(num_obstacles, 2, generator=g)
            if dense:
                ctr = ctr * 0.2
            for b in range(o, o + a):
                Wm = world_from_agent[b]
                wx = Wm[0, 0] * ax + Wm[0, 1] * ay + Wm[0, 2]
                wy = Wm[1, 0] * ax + Wm[1, 1] * ay + Wm[1, 2]
                for q in range(num_obstacles):
                    inside = ((wx - ctr[q, 0]).abs() < half[q, 0]) & ((wy - ctr[q, 1]).abs() < half[q, 1])
                    drivable[b] &= ~inside
            o += a
        image = torch.zeros(B, C, H, W)
        image[:, 0] = drivable.float()
        if C > 1:
            image[:, 1] = 1.0 - drivable.float()
        batch["drivable_map"] = drivable
        batch["image"] = image

    # stand-in for the map encoder's output
    gg = torch.Generator().manual_seed(seed + 7919 if grid_seed is None else grid_seed)
    Hg, Wg = cfg.grid_hw
    batch["map_grid_feat"] = torch.randn(B, cfg.map_grid_feature_dim, Hg, Wg, generator=gg)
    nc = torch.randn(1, cfg.map_grid_feature_dim, Hg, Wg, generator=gg)
    batch["map_grid_feat_non_cond"] = nc.repeat(B, 1, 1, 1)
    return batch


def make_noise(cfg: ModelConfig, num_agents, num_samp, seed=0):
    """Noise tape in the reference's consumption order:
    init (B, N, T, 6), steps (n_steps, B*N, T, 2); steps[k] is used at t = n_steps-1-k."""
    g = torch.Generator().manual_seed(seed)
    init = torch.randn(num_agents, num_samp, cfg.horizon, 6, generator=g)
    steps = torch.randn(cfg.n_diffusion_steps, num_agents * num_samp, cfg.horizon, 2, generator=g)
    return init, steps


def target_pos_guidance(batch, agents_per_scene, weight=30000.0, seed=0, ahead=4.0):
    """configs/eval/orca/target_pos.json:158-166 shaped guidance: one agent-frame
    waypoint ~``ahead`` m in front of each agent."""
    g = torch.Generator().manual_seed(seed + 13)
    out = []
    for a in agents_per_scene:
        tgt = torch.tensor([ahead, 0.0])[None] + torch.randn(a, 2, generator=g)
        out.append([{"name": "target_pos", "weight": weight,
                     "params": {"target_pos": tgt.tolist()}, "agents": None}])
    return out


def collision_guidance(agents_per_scene, agent_weight=1000.0, map_weight=10.0, num_disks=1,
                       buffer_dist=0.2, num_points_lw=(10, 10)):
    """configs/eval/orca/target_pos_all_coll.json:158-176 shaped guidance."""
    return [[{"name": "agent_collision", "weight": agent_weight,
              "params": {"num_disks": num_disks, "buffer_dist": buffer_dist}, "agents": None},
             {"name": "map_collision", "weight": map_weight,
              "params": {"num_points_lw": list(num_points_lw)}, "agents": None}]
            for _ in agents_per_scene]


def merge_guidance(*cfgs):
    """Per-scene concatenation (reference tbsim/utils/scene_edit_utils.py:117-126)."""
    n = len(cfgs[0])
    return [sum((c[i] for c in cfgs), []) for i in range(n)]
