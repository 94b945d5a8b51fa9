"""Host side of test-time guidance: parses the reference's guidance config lists
(list over scenes of ``{name, weight, params, agents}`` dicts,
/root/reference/tbsim/utils/guidance_loss.py:70-105,866-921) into per-agent tables and
launches the HIP objective kernels (include/trace_hip.h, ``trace_guide_*``).

The reference loops ``for scene: for guide:`` in Python every diffusion step and builds a
(B, N) NaN-padded tensor per guide; here configs of the same kind with disjoint agent masks
are merged into one launch ("layer") and per-guide losses are only materialised on request.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _cabi
from ._cabi import ptr

GUIDANCE_NAMES = ("target_speed", "agent_collision", "map_collision", "target_pos_at_time",
                  "target_pos", "global_target_pos_at_time", "global_target_pos", "min_speed",
                  "social_group")
_UNSUPPORTED = {"amp_value": "needs an external value function (guidance_loss.py:816-849)"}
SOCIAL_GROUP_WIDTH = 25          # TRACE_SOCIAL_GROUP_WIDTH of include/trace_hip.h: up to here the exact distance form
SCENE_LEVEL = ("agent_collision", "social_group")


class GuidanceConfig(object):
    """Same fields / checks as the reference's GuidanceConfig (guidance_loss.py:70-93)."""

    def __init__(self, name, weight, params, agents, func=None):
        if name in _UNSUPPORTED:
            raise NotImplementedError("guidance '%s' is not built: %s" % (name, _UNSUPPORTED[name]))
        assert name in GUIDANCE_NAMES, \
            "Guidance name must be one of: " + ", ".join(GUIDANCE_NAMES)
        self.name, self.weight, self.params, self.agents, self.func = name, weight, params, agents, func

    @staticmethod
    def from_dict(config_dict):
        assert config_dict.keys() == {"name", "weight", "params", "agents"}, \
            "Guidance config must include only [name, weight, params, agt_mask]. " \
            "agt_mask may be None if applies to all agents in a scene"
        return GuidanceConfig(**config_dict)

    def __repr__(self):
        return "<\n%s\n>" % str("\n ".join("%s : %s" % (k, repr(v)) for (k, v) in self.__dict__.items()))


def verify_guidance_config_list(guidance_config_list):
    """guidance_loss.py:95-105."""
    assert len(guidance_config_list) > 0
    return any(len(g) > 0 for g in guidance_config_list)


def choose_action_from_guidance(preds, obs_dict, guide_configs, guide_losses):
    """Best-of-N sample per agent from the per-guide losses, the reference's public free function
    (guidance_loss.py:15-41): ``guide_losses`` maps "<name>_scene_%03d_%02d" -> (B, N) with NaN outside the
    guide's agents, in scene-major order; ``guide_configs`` is the per-scene list of configs (objects with
    ``.name`` or the config dicts).  Unweighted nansum over a scene's guides; one argmin per scene when the
    scene has an agent_collision / social_group guide, per agent otherwise; agents no guide covers keep
    sample 0.  Stays on the device (the reference copies to the host every call).  The sampler's own path
    (``GuidancePlan.choose_action``) works on the compact per-layer losses instead."""
    pos = preds["positions"]
    B, N = pos.shape[0], pos.shape[1]
    act_idx = torch.zeros(B, dtype=torch.long, device=pos.device)
    per_guide = [v for _, v in guide_losses.items()]
    first = 0
    for scene_cfgs in guide_configs:
        mine = per_guide[first:first + len(scene_cfgs)]
        first += len(scene_cfgs)
        stack = torch.stack(mine, dim=2) if mine else pos.new_zeros(B, N, 0)
        covered = ~torch.isnan(stack.sum(dim=(1, 2)))       # an empty guide list covers everybody (reference quirk)
        if not bool(covered.any()):
            continue
        total = torch.nansum(stack[covered], dim=-1)        # (B', N)
        names = [c["name"] if isinstance(c, dict) else c.name for c in scene_cfgs]
        if any(n in SCENE_LEVEL for n in names):
            act_idx[covered] = torch.argmin(total.sum(dim=0))
        else:
            act_idx[covered] = torch.argmin(total, dim=-1)
    return act_idx


class _Item:
    """One kernel-level objective over a set of agents (part of one guidance config)."""
    __slots__ = ("kind", "key", "mask", "tables", "layer")

    def __init__(self, kind, key, mask, tables):
        self.kind, self.key, self.mask, self.tables, self.layer = kind, key, mask, tables, None


class _Instance:
    """One guidance config of one scene: loss = sum of its items (disjoint sub-masks)."""
    __slots__ = ("name", "key_name", "scene", "mask", "items")

    def __init__(self, name, key_name, scene, mask):
        self.name, self.key_name, self.scene, self.mask, self.items = name, key_name, scene, mask, []


class GuideLosses(OrderedDict):
    """Mapping "<name>_scene_%03d_%02d" -> (B, N) loss with NaN outside the guide's agent mask
    (guidance_loss.py:916-918), materialised lazily from the compact per-layer losses."""

    def __init__(self, plan, layer_losses):
        super().__init__()
        self._plan = plan
        self._layer_losses = layer_losses
        for inst in plan.instances:
            OrderedDict.__setitem__(self, inst.key_name, None)

    def __getitem__(self, k):
        v = OrderedDict.__getitem__(self, k)
        if v is None:
            v = self._plan.instance_loss(k, self._layer_losses)
            OrderedDict.__setitem__(self, k, v)
        return v

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class DiffuserGuidance(object):
    """Counterpart of the reference's DiffuserGuidance (guidance_loss.py:866-928)."""

    def __init__(self, guidance_config_list, example_batch=None):
        self.num_scenes = len(guidance_config_list)
        assert self.num_scenes > 0, "Guidance config list must include list of guidance for each scene"
        self.guide_configs = [[]] * self.num_scenes
        for si in range(self.num_scenes):
            if len(guidance_config_list[si]) > 0:
                self.guide_configs[si] = [GuidanceConfig.from_dict(c) for c in guidance_config_list[si]]
        self.global_t = 0
        # social_group only: injected neighbour draws {guide key name: (randint, rand)}, reused at every
        # evaluation (parity tests); None = drawn on the device per evaluation
        self.social_tape = None

    def update(self, global_t=None, **kwargs):
        if global_t is not None:
            self.global_t = global_t

    def build_plan(self, data_batch, num_samp, horizon, device, generator=None):
        plan = GuidancePlan(self, data_batch, num_samp, horizon, device)
        plan.social_tape, plan.generator = self.social_tape, generator
        return plan


def _transform_points(pts, mat):
    # geometry_utils.py:105-171 for (B,1,2) points and (B,3,3) matrices
    mt = mat.transpose(-1, -2)
    return pts @ mt[:, :2, :2] + mt[:, -1:, :2]


class GuidancePlan:
    """Per-sampling-call tables (they depend on the batch: extents, transforms, masks)."""

    def __init__(self, guidance, batch, num_samp, horizon, device):
        self.N, self.T, self.device = int(num_samp), int(horizon), device
        scene_index = batch["scene_index"].detach().cpu()
        B = scene_index.shape[0]
        self.B = B
        _, sid = torch.unique_consecutive(scene_index, return_inverse=True)
        counts = torch.bincount(sid)
        starts = torch.cumsum(counts, 0) - counts
        self.scene_id = sid
        self.scene_start_t = starts[sid].to(torch.int32)
        self.scene_size_t = counts[sid].to(torch.int32)
        cpu = {k: batch[k].detach().cpu() for k in ("extent", "agent_from_world") if k in batch}
        self.instances = []
        items = []
        for si in range(guidance.num_scenes):
            for gi, gc in enumerate(guidance.guide_configs[si]):
                mask = sid == si
                if gc.agents is not None:
                    sel = torch.nonzero(mask, as_tuple=True)[0][gc.agents]
                    mask = torch.zeros_like(mask)
                    mask[sel] = True
                nm = int(mask.sum())
                inst = _Instance(gc.name, "%s_scene_%03d_%02d" % (gc.name, si, gi), si, mask)
                scale = float(gc.weight) / float(max(nm, 1) * self.N)
                inst.items = self._make_items(gc, mask, nm, scale, cpu, guidance.global_t)
                items += inst.items
                self.instances.append(inst)
        # pack items into layers: same kind/static key, disjoint masks
        self.layers = []
        for it in items:
            # scene-coupled objectives: two configs of ONE scene never share a launch -- the pair kernel doubles
            # a pair's gradient when both agents are guided by the SAME config, while the reference detaches the
            # agents of any other config (guidance_loss.py:272-282)
            claim = it.mask
            if it.kind in SCENE_LEVEL:
                claim = sid == int(sid[it.mask][0]) if bool(it.mask.any()) else it.mask
            for li, ly in enumerate(self.layers):
                if ly["kind"] == it.kind and ly["key"] == it.key and not bool((ly["claim"] & claim).any()):
                    break
            else:
                li = len(self.layers)
                self.layers.append({"kind": it.kind, "key": it.key, "mask": torch.zeros(B, dtype=torch.bool),
                                    "claim": torch.zeros(B, dtype=torch.bool), "tables": {}})
                ly = self.layers[li]
            it.layer = li
            ly["mask"] |= it.mask
            ly["claim"] |= claim
            for k, v in it.tables.items():
                if k not in ly["tables"]:
                    # scale < 0 marks agents the layer does not cover; a covered agent of weight 0 keeps its
                    # loss (the reference still evaluates and reports it, and selection uses it)
                    fill = -1.0 if k == "scale" else 0
                    ly["tables"][k] = torch.full((B,) + tuple(v.shape[1:]), fill, dtype=v.dtype)
                ly["tables"][k][it.mask] = v
        # device tables
        f32 = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
        self.extent = f32(batch["extent"])
        self.world_from_agent = f32(batch["world_from_agent"]) if "world_from_agent" in batch else None
        self.raster_from_agent = f32(batch["raster_from_agent"]) if "raster_from_agent" in batch else None
        self.drivable = None
        self.scene_start = self.scene_start_t.to(device)
        self.scene_size = self.scene_size_t.to(device)
        for ly in self.layers:
            ly["dev"] = {k: v.to(device).contiguous() for k, v in ly["tables"].items()}
            if ly["kind"] == "agent_collision":
                D = ly["key"][1]
                ext = cpu["extent"]
                rad = ext[:, 1] / 2.0
                lo = -(ext[:, 0] / 2.0) + rad
                hi = (ext[:, 0] / 2.0) - rad
                dx = torch.stack([torch.linspace(lo[i].item(), hi[i].item(), D) for i in range(B)], 0)
                ly["dev"]["radius"] = f32(rad)
                ly["dev"]["disk_x"] = f32(dx)
                ly["dev"]["in_mask"] = ly["mask"].to(torch.int32).to(device)
                assert self.world_from_agent is not None, "agent_collision needs world_from_agent"
            if ly["kind"] == "map_collision":
                nl, nw = ly["key"][1]
                loc = torch.cartesian_prod(torch.linspace(-0.5, 0.5, nl), torch.linspace(-0.5, 0.5, nw))
                ly["dev"]["local"] = f32(loc)
                if self.drivable is None:
                    self.drivable = batch["drivable_map"].to(device=device, dtype=torch.uint8).contiguous()
        M = B * self.N
        self.ws = torch.empty(M * self.T * 4, dtype=torch.float32, device=device)
        self.debug_map_argmin = False    # tests: evaluate() records the map-collision arg-min choices in map_argmin
        self.map_argmin = {}
        self.generator = None            # torch.Generator for the social_group neighbour draws
        self.social_tape = None          # {guide key name: (randint (T*N, G, 1), rand (T*N, G))}: injected draws
        if any(ly["kind"] == "social_group" for ly in self.layers):
            assert self.world_from_agent is not None, "social_group needs world_from_agent"
            self.nbr_ws = torch.empty(M * self.T, dtype=torch.int32, device=device)
            self.dist_ws = torch.empty(M * self.T, dtype=torch.float32, device=device)
        # selection tables (choose_action_from_guidance, guidance_loss.py:15-41)
        n_guides = torch.tensor([len(g) for g in guidance.guide_configs], dtype=torch.long)
        cover = torch.zeros(B, dtype=torch.long)
        for inst in self.instances:
            cover += inst.mask.long()
        valid = (cover == n_guides[sid]) & (n_guides[sid] > 0)
        # Reference behaviour kept on purpose: for a scene with an EMPTY guide list the slice of
        # losses is zero-width, so its NaN test passes for every agent of the batch and
        # act_idx[all] = 0 (guidance_loss.py:26-39) -- the choices of all earlier scenes are reset.
        empty = torch.nonzero(n_guides == 0, as_tuple=True)[0]
        if empty.numel() > 0:
            valid &= sid > int(empty.max())
        self.sel_valid = valid.to(device)
        scene_level = torch.tensor([any(c.name in SCENE_LEVEL for c in g) for g in guidance.guide_configs])
        self.sel_scene_level = scene_level[sid].to(device)
        self.sel_scene_id = sid.to(device)
        self.sel_num_scenes = int(sid.max().item()) + 1 if B > 0 else 0

    # -- hipGraph replay support ------------------------------------------------------ #
    _SHARED = ("extent", "world_from_agent", "raster_from_agent", "drivable", "scene_start",
               "scene_size", "sel_valid", "sel_scene_level", "sel_scene_id")

    def signature(self):
        """Hashable description of everything a captured launch sequence depends on besides the
        CONTENTS of the device tables: layer kinds / static keys / masks and table shapes."""
        if any(ly["kind"] == "social_group" for ly in self.layers):
            raise NotImplementedError("social_group draws random neighbours every evaluation: the step loop "
                                      "cannot be replayed as a captured graph (use_graph=False)")
        layers = tuple((ly["kind"], ly["key"], tuple(ly["mask"].tolist()),
                        tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in ly["dev"].items())))
                       for ly in self.layers)
        shared = tuple((n, None if getattr(self, n) is None else tuple(getattr(self, n).shape))
                       for n in self._SHARED)
        return (self.B, self.N, self.T, layers, shared, tuple(i.key_name for i in self.instances),
                self.sel_num_scenes)

    def load_tables_from(self, other):
        """Copies the device tables of `other` (same signature) into this plan's tensors, whose
        addresses a captured graph holds."""
        assert self.signature() == other.signature(), "guidance structure changed: re-capture"
        for n in self._SHARED:
            t = getattr(self, n)
            if t is not None:
                t.copy_(getattr(other, n))
        for a, b in zip(self.layers, other.layers):
            for k, v in a["dev"].items():
                v.copy_(b["dev"][k])

    # -- config -> kernel items ------------------------------------------------------ #
    def _make_items(self, gc, mask, nm, scale, cpu, global_t):
        T = self.T
        p = gc.params
        sc = torch.full((nm,), scale, dtype=torch.float32)
        name = gc.name
        if name == "target_pos":
            tgt = torch.as_tensor(p["target_pos"], dtype=torch.float32).reshape(-1, 2)
            assert nm == tgt.shape[0]
            t0 = int(p.get("min_target_time", 0.0) * T)
            return [_Item("target_pos", ("target_pos",), mask,
                          {"target": tgt, "t_min": torch.full((nm,), t0, dtype=torch.int32), "scale": sc})]
        if name == "target_pos_at_time":
            tgt = torch.as_tensor(p["target_pos"], dtype=torch.float32).reshape(-1, 2)
            tt = torch.as_tensor(p["target_time"]).to(torch.int32).reshape(-1)
            assert nm == tgt.shape[0] and nm == tt.shape[0]
            assert bool(((tt >= 0) & (tt < T)).all()), "target_time outside the planning horizon"
            return [_Item("target_pos_at_time", ("target_pos_at_time",), mask,
                          {"target": tgt, "t_at": tt, "scale": sc})]
        if name in ("target_speed", "min_speed"):
            if name == "min_speed":
                kind, s = 2, p["min_speed"]
            else:
                mode = p.get("mode", "use_action")
                assert mode in ("use_action", "use_position")
                kind, s = (0 if mode == "use_action" else 1), p["target_speed"]
                if isinstance(s, (list, tuple)):
                    # the reference builds a (B',) tensor and broadcasts it against (B', N, T):
                    # only the one-agent form broadcasts as intended (guidance_loss.py:162-169)
                    if len(s) != 1 or nm != 1:
                        raise ValueError("target_speed as a per-agent list does not broadcast in the "
                                         "reference (guidance_loss.py:162-169); pass a scalar")
                    s = s[0]
            return [_Item("speed", ("speed", kind, float(p["dt"])), mask,
                          {"speed": torch.full((nm,), float(s), dtype=torch.float32), "scale": sc})]
        if name == "agent_collision":
            return [_Item("agent_collision",
                          ("agent_collision", int(p.get("num_disks", 5)), float(p.get("buffer_dist", 0.2))),
                          mask, {"scale": sc})]
        if name == "map_collision":
            nlw = tuple(int(v) for v in p.get("num_points_lw", (10, 10)))
            return [_Item("map_collision", ("map_collision", nlw), mask, {"scale": sc})]
        if name == "social_group":
            coh = float(p.get("cohesion", 0.8))
            assert 0.0 <= coh <= 1.0
            if nm < 2:
                raise ValueError("social_group needs at least two agents (the reference draws "
                                 "torch.randint(0, B-1), guidance_loss.py:803)")
            # members table: 25 wide (every group the exact distance form covers), or the group's size rounded up to
            # a multiple of 8 (the kernel switches to torch.cdist's matmul form by the member count, as ATen does);
            # the width is part of the layer key, so tables of different widths never share a launch
            width = SOCIAL_GROUP_WIDTH if nm <= SOCIAL_GROUP_WIDTH else (nm + 7) // 8 * 8
            idx = torch.nonzero(mask, as_tuple=True)[0]                    # group order = batch order
            members = torch.full((nm, width), -1, dtype=torch.int32)
            members[:, :nm] = idx.to(torch.int32)[None]
            # the reference compares the BATCH index with leader_idx (guidance_loss.py:775): kept
            leader = (idx == int(p.get("leader_idx", 0))).to(torch.int32)
            return [_Item("social_group", ("social_group", float(p.get("social_dist", 1.5)), width), mask,
                          {"scale": sc, "members": members, "is_leader": leader,
                           "cohesion": torch.full((nm,), coh, dtype=torch.float32),
                           "gsize": torch.full((nm,), nm, dtype=torch.int64),
                           "slot": torch.arange(nm, dtype=torch.int64)})]
        if name in ("global_target_pos", "global_target_pos_at_time"):
            tgt_w = torch.tensor(p["target_pos"], dtype=torch.float32).reshape(-1, 2)
            urg = torch.tensor(p["urgency"], dtype=torch.float32).reshape(-1)
            assert nm == tgt_w.shape[0] and nm == urg.shape[0]
            afw = cpu["agent_from_world"][mask]
            local = _transform_points(tgt_w[:, None], afw)[:, 0]
            pref, dt = p.get("pref_speed", 1.42), p.get("dt", 0.1)
            idx = torch.nonzero(mask, as_tuple=True)[0]
            out = []

            def sub(sel):
                mk = torch.zeros_like(mask)
                mk[idx[sel]] = True
                return mk

            if name == "global_target_pos":
                exact = torch.norm(local, dim=-1) < T * dt * pref
                prog = ~exact
                goal = torch.maximum(urg * (T * dt * pref),
                                     torch.tensor([p.get("min_progress_dist", 0.5)]))
                if bool(prog.any()):
                    k = int(prog.sum())
                    out.append(_Item("progress", ("progress",), sub(prog),
                                     {"target": local[prog], "goal": goal[prog],
                                      "mode": torch.ones(k, dtype=torch.int32), "scale": sc[:k]}))
                if bool(exact.any()):
                    k = int(exact.sum())
                    out.append(_Item("target_pos", ("target_pos",), sub(exact),
                                     {"target": local[exact], "t_min": torch.zeros(k, dtype=torch.int32),
                                      "scale": sc[:k]}))
            else:
                tt = torch.tensor(p["target_time"]).reshape(-1)
                assert nm == tt.shape[0]
                lt = tt - global_t
                exact = (lt < T) & (lt >= 0)
                prog = (~exact) & (lt >= 0)
                if bool(prog.any()):
                    k = int(prog.sum())
                    goal = lt[prog] * dt * pref * (1.0 - urg[prog])
                    out.append(_Item("progress", ("progress",), sub(prog),
                                     {"target": local[prog], "goal": goal.to(torch.float32),
                                      "mode": torch.zeros(k, dtype=torch.int32), "scale": sc[:k]}))
                if bool(exact.any()):
                    k = int(exact.sum())
                    out.append(_Item("target_pos_at_time", ("target_pos_at_time",), sub(exact),
                                     {"target": local[exact], "t_at": lt[exact].to(torch.int32),
                                      "scale": sc[:k]}))
            return out
        raise AssertionError(name)

    # -- evaluation --------------------------------------------------------------------- #
    def evaluate(self, traj, grad_traj, stream):
        """traj (M,T,6) descaled; grad_traj (M,T,6) zeroed by the caller or None.
        Returns the list of per-layer loss tensors (M,) (zeros where a layer is inactive)."""
        L = _cabi.lib()
        M, N, T = self.B * self.N, self.N, self.T
        gp = ptr(grad_traj)
        out = []
        for ly in self.layers:
            d = ly["dev"]
            loss = torch.zeros(M, dtype=torch.float32, device=self.device)
            kind = ly["kind"]
            if kind == "target_pos":
                rc = L.trace_guide_target_pos(ptr(traj), gp, ptr(loss), ptr(d["target"]), ptr(d["t_min"]),
                                              ptr(d["scale"]), M, N, T, stream)
            elif kind == "target_pos_at_time":
                rc = L.trace_guide_target_pos_at_time(ptr(traj), gp, ptr(loss), ptr(d["target"]),
                                                      ptr(d["t_at"]), ptr(d["scale"]), M, N, T, stream)
            elif kind == "progress":
                rc = L.trace_guide_progress(ptr(traj), gp, ptr(loss), ptr(d["target"]), ptr(d["goal"]),
                                            ptr(d["mode"]), ptr(d["scale"]), M, N, T, stream)
            elif kind == "speed":
                rc = L.trace_guide_speed(ptr(traj), gp, ptr(loss), ptr(d["speed"]), ptr(d["scale"]),
                                         ly["key"][1], ly["key"][2], M, N, T, stream)
            elif kind == "agent_collision":
                rc = L.trace_guide_agent_collision(
                    ptr(traj), gp, ptr(loss), ptr(self.world_from_agent), ptr(d["radius"]),
                    ptr(d["disk_x"]), ly["key"][1], ly["key"][2], ptr(self.scene_start),
                    ptr(self.scene_size), ptr(d["in_mask"]), ptr(d["scale"]), self.B, N, T,
                    ptr(self.ws), stream)
            elif kind == "map_collision":
                H, W = self.drivable.shape[-2:]
                if self.debug_map_argmin and gp is not None:      # the evaluation that produces the gradient
                    # tests only: also record which on-road point every off-road point was paired with
                    P = d["local"].shape[0]
                    am = torch.empty(M * T, P, 2, dtype=torch.int32, device=self.device)
                    rc = L.trace_guide_map_collision_argmin(
                        ptr(traj), gp, ptr(loss), ptr(self.drivable), ptr(self.extent),
                        ptr(self.raster_from_agent), ptr(d["local"]), P, ptr(d["scale"]),
                        H, W, M, N, T, ptr(self.ws), ptr(am), stream)
                    self.map_argmin[ly["key"]] = am
                else:
                    rc = L.trace_guide_map_collision(
                        ptr(traj), gp, ptr(loss), ptr(self.drivable), ptr(self.extent),
                        ptr(self.raster_from_agent), ptr(d["local"]), d["local"].shape[0], ptr(d["scale"]),
                        H, W, M, N, T, ptr(self.ws), stream)
            elif kind == "social_group":
                pick = self._social_pick(ly)
                rc = L.trace_guide_social_group(
                    ptr(traj), gp, ptr(loss), ptr(self.world_from_agent), ptr(d["members"]),
                    int(d["members"].shape[1]), ptr(d["is_leader"]), ptr(pick), ly["key"][1], ptr(d["scale"]),
                    self.B, N, T, ptr(self.ws), ptr(self.nbr_ws), ptr(self.dist_ws), stream)
            else:
                raise AssertionError(kind)
            _cabi.check(rc, "trace_guide_" + kind)
            out.append(loss)
        return out

    @staticmethod
    def decode_map_argmin(am):
        """(M*T, P, 2) int32 of trace_guide_map_collision_argmin -> ((M*T, P, 4) int64 tied rows, -1 padded;
        (M*T, P) number of tied rows, 0 where the point is not an off-road point of an active step)."""
        am = am.cpu().long()
        a, b = am[..., 0], am[..., 1] & 0xFFFFFF
        nt = torch.where(a >= 0, a >> 8, torch.zeros_like(a))
        rows = torch.stack([a & 255, b & 255, (b >> 8) & 255, (b >> 16) & 255], -1)
        k = torch.arange(4).expand_as(rows)
        rows = torch.where((k < nt.unsqueeze(-1).clamp(max=4)) & (a >= 0).unsqueeze(-1), rows, torch.full_like(rows, -1))
        return rows, nt

    def _social_pick(self, ly):
        """(B, N, T) int32: the randomly assigned neighbour (global agent index) or -1 = nearest.
        guidance_loss.py:797-808: with probability `cohesion` a member is paired with a uniformly drawn
        OTHER member of its group instead of its nearest one.  The reference draws on the host every
        evaluation; here the draw is on the device (``self.generator``), or injected (``social_tape``)."""
        d, dev = ly["dev"], self.device
        B, N, T = self.B, self.N, self.T
        gs, slot = d["gsize"], d["slot"]
        if self.social_tape is not None:
            r = torch.zeros(B, N, T, dtype=torch.int64, device=dev)
            u = torch.ones(B, N, T, dtype=torch.float32, device=dev)
            for inst in self.instances:
                if inst.name == "social_group" and inst.items[0].layer == self.layers.index(ly):
                    ri, ui = self.social_tape[inst.key_name]
                    G = int(inst.mask.sum())
                    mk = inst.mask.to(dev)
                    r[mk] = ri.to(dev).reshape(T, N, G).permute(2, 1, 0).to(torch.int64)
                    u[mk] = ui.to(dev).reshape(T, N, G).permute(2, 1, 0).to(torch.float32)
        else:
            span = (gs - 1).clamp(min=1).to(torch.float32)[:, None, None]
            r = (torch.rand(B, N, T, device=dev, generator=self.generator) * span).to(torch.int64)
            r = torch.minimum(r, (gs - 2).clamp(min=0)[:, None, None])
            u = torch.rand(B, N, T, device=dev, generator=self.generator)
        rs = r + (r >= slot[:, None, None]).to(torch.int64)             # skip self in the choice list
        cand = torch.gather(d["members"].to(torch.int64), 1,
                            rs.reshape(B, N * T).clamp(max=d["members"].shape[1] - 1)).reshape(B, N, T)
        take = u < d["cohesion"][:, None, None]
        return torch.where(take, cand, torch.full_like(cand, -1)).to(torch.int32).contiguous()

    def instance_loss(self, key_name, layer_losses):
        for inst in self.instances:
            if inst.key_name == key_name:
                full = torch.full((self.B, self.N), float("nan"), device=self.device)
                # agents of the mask not covered by an item (e.g. passed at-time targets) are 0
                m = inst.mask.to(self.device)
                full[m] = 0.0
                for it in inst.items:
                    mk = it.mask.to(self.device)
                    full[mk] = layer_losses[it.layer].view(self.B, self.N)[mk]
                return full
        raise KeyError(key_name)

    def guide_losses(self, layer_losses):
        return GuideLosses(self, layer_losses)

    def total_scores(self, layer_losses):
        """(B, N) unweighted sum of all guidance losses covering each agent."""
        B, N, dev = self.B, self.N, self.device
        total = torch.zeros(B, N, device=dev)
        for ly, l in zip(self.layers, layer_losses):
            total += l.view(B, N) * ly["mask"].to(dev).unsqueeze(1)
        return total

    def choose_action(self, layer_losses):
        """choose_action_from_guidance (guidance_loss.py:15-41), on device and without the
        per-guide NaN-padded tensors: unweighted sum of the guides per agent; per-agent argmin,
        or one argmin per scene when the scene has a scene-level guide."""
        B, N, dev = self.B, self.N, self.device
        total = self.total_scores(layer_losses)
        idx = torch.zeros(B, dtype=torch.long, device=dev)
        valid = self.sel_valid
        per_agent = torch.argmin(total, dim=-1)
        scene_tot = torch.zeros(self.sel_num_scenes, N, device=dev)
        scene_tot.index_add_(0, self.sel_scene_id[valid], total[valid])
        per_scene = torch.argmin(scene_tot, dim=-1)[self.sel_scene_id]
        choice = torch.where(self.sel_scene_level, per_scene, per_agent)
        return torch.where(valid, choice, idx)
