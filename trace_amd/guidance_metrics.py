"""Per-guidance success metrics of a closed-loop rollout: the `guide_<metric>_s<scene>g<config>` entries the
reference adds to the environment's metrics while a guidance configuration is active and reports in `stats.json`.

Behaviour restated from /root/reference/tbsim/utils/guidance_metrics.py (class by class, lines cited below) and wired
in as /root/reference/tbsim/utils/scene_edit_utils.py:47-60 does (`guidance_metrics_from_config` -> `env._metrics`).
Every metric handles ONLY the scene (local index inside the batch) and the agents of its guidance config, is fed the
state BEFORE each simulated action (env_trajdata.py:345-358: the first state it sees is the one the plan was made
from), and returns one value per scene of the batch, NaN everywhere but its own scene.

Differences by construction: states arrive as torch tensors on the environment's device (the reference converts to
numpy per step); the oriented-box collision test is a separating-axis test (the reference asks shapely whether the
two polygons intersect -- the same predicate up to boundary contact); rendering / data-frame hooks of `EnvMetrics`
are not part of this path.
"""
import math

import numpy as np
import torch


def _local_scene_index(state_info):
    """Index of each agent's scene inside the batch (guidance_metrics.py:50: unique_consecutive ... inverse)."""
    sid = torch.as_tensor(state_info["scene_index"])
    return torch.unique_consecutive(sid, return_inverse=True)[1]


def _t(x, like=None):
    x = torch.as_tensor(x)
    return x if like is None else x.to(like.device)


class GuidanceMetric:
    """guidance_metrics.py:22-38."""

    def __init__(self, scene_idx, params, agents, num_scenes):
        self.scene_idx = int(scene_idx)
        self.params = params
        self.agents = None if agents is None else [int(a) for a in agents]
        self.num_scenes = int(num_scenes)
        self.reset()

    # the environment's metric protocol (env_metrics.EnvMetrics)
    def reset(self):
        raise NotImplementedError

    def multi_episode_reset(self):
        self.reset()

    def _select(self, state_info, *keys):
        """The rows of this metric's scene, then of its agents, of the named per-agent arrays."""
        mask = _local_scene_index(state_info) == self.scene_idx
        out = []
        for k in keys:
            v = _t(state_info[k])
            v = v[mask.to(v.device)]
            if self.agents is not None:
                v = v[torch.as_tensor(self.agents, device=v.device)]
            out.append(v)
        return out if len(out) > 1 else out[0]

    def _scene_only(self, state_info, key):
        v = _t(state_info[key])
        return v[(_local_scene_index(state_info) == self.scene_idx).to(v.device)]

    def _for_scene(self, value):
        met = np.ones(self.num_scenes) * np.nan
        met[self.scene_idx] = float(value)
        return met


class TargetSpeedGuidance(GuidanceMetric):
    """Mean |speed - target_speed| of the guided agents over the episode (guidance_metrics.py:40-69)."""

    def reset(self):
        self._per_step = []

    def add_step(self, state_info, all_scene_index):
        speed = self._select(state_info, "curr_speed").float()
        tgt = torch.as_tensor(self.params["target_speed"], dtype=torch.float32, device=speed.device)
        self._per_step.append(float((speed - tgt).abs().mean()))

    def get_episode_metrics(self):
        return self._for_scene(np.mean(self._per_step))


def _box_corners(pos, yaw, extent):
    """(n, 4, 2) corners of oriented boxes: extent = (length, width) along / across the heading."""
    c, s = torch.cos(yaw), torch.sin(yaw)
    hl, hw = extent[:, 0] / 2.0, extent[:, 1] / 2.0
    dx = torch.stack([hl, hl, -hl, -hl], -1)
    dy = torch.stack([hw, -hw, -hw, hw], -1)
    x = pos[:, 0:1] + dx * c[:, None] - dy * s[:, None]
    y = pos[:, 1:2] + dx * s[:, None] + dy * c[:, None]
    return torch.stack([x, y], -1)


def boxes_intersect(pos, yaw, extent):
    """(n, n) bool: do the oriented boxes of agents i and j overlap (separating-axis test on the four edge normals)."""
    n = pos.shape[0]
    cor = _box_corners(pos, yaw, extent)                                   # (n, 4, 2)
    ax = torch.stack([torch.stack([torch.cos(yaw), torch.sin(yaw)], -1),
                      torch.stack([-torch.sin(yaw), torch.cos(yaw)], -1)], 1)   # (n, 2 axes, 2)
    hit = torch.ones(n, n, dtype=torch.bool, device=pos.device)
    for axes in (ax[:, None].expand(n, n, 2, 2), ax[None].expand(n, n, 2, 2)):   # axes of box i, then of box j
        pi = torch.einsum("ijad,ikd->ijak", axes, cor)                     # projections of box i's corners
        pj = torch.einsum("ijad,jkd->ijak", axes, cor)                     # ... of box j's corners
        sep = (pi.max(-1).values < pj.min(-1).values) | (pj.max(-1).values < pi.min(-1).values)
        hit &= ~sep.any(-1)
    hit &= ~torch.eye(n, dtype=torch.bool, device=pos.device)
    return hit


class AgentCollisionGuidance(GuidanceMetric):
    """Fraction of the guided agents whose box ever overlapped another agent of the scene
    (guidance_metrics.py:71-122)."""

    def reset(self):
        self._per_step = []

    def _coll(self, state_info):
        pos = self._scene_only(state_info, "centroid").float()
        yaw = self._scene_only(state_info, "yaw").float().reshape(-1)
        ext = self._scene_only(state_info, "extent").float()[..., :2]
        return boxes_intersect(pos, yaw, ext).any(-1).float()

    def add_step(self, state_info, all_scene_index):
        coll = self._coll(state_info)
        if self.agents is not None:
            coll = coll[torch.as_tensor(self.agents, device=coll.device)]
        self._per_step.append(coll)

    def get_episode_metrics(self):
        allc = torch.stack(self._per_step, 1)                              # (agents, steps)
        return self._for_scene(allc.amax(1).mean())


class AgentCollisionGuidanceDisk(AgentCollisionGuidance):
    """The same with one disk per agent (radius = min extent / 2), optionally widened by the guidance's
    buffer distance (guidance_metrics.py:124-182)."""

    def __init__(self, scene_idx, params, agents, num_scenes, use_buffer_dist=False):
        self.use_buffer_dist = use_buffer_dist
        super().__init__(scene_idx, params, agents, num_scenes)

    def _coll(self, state_info):
        pos = self._scene_only(state_info, "centroid").float()
        ext = self._scene_only(state_info, "extent").float()[..., :2]
        rad = ext.amin(-1) / 2.0
        d = torch.linalg.norm(pos[:, None] - pos[None], dim=-1)
        allowed = rad[:, None] + rad[None]
        if self.use_buffer_dist:
            allowed = allowed + float(self.params["buffer_dist"])
        hit = (d < allowed) & ~torch.eye(pos.shape[0], dtype=torch.bool, device=pos.device)
        return hit.any(-1).float()


class SocialDistanceGuidance(AgentCollisionGuidanceDisk):
    def __init__(self, scene_idx, params, agents, num_scenes):
        super().__init__(scene_idx, params, agents, num_scenes, use_buffer_dist=True)


def _transform(points, mat):
    """(n, 2) points through (n, 3, 3) homogeneous transforms."""
    return torch.einsum("nij,nj->ni", mat[:, :2, :2], points) + mat[:, :2, 2]


class TargetPosGuidance(GuidanceMetric):
    """Mean over the guided agents of the smallest distance to their target, measured in the agent frame of
    the FIRST state seen (the planning frame the target was given in) (guidance_metrics.py:184-218)."""
    _local_frame = True

    def reset(self):
        self.locs = torch.as_tensor(np.array(self.params["target_pos"]), dtype=torch.float32)
        self._afw = None
        self._min = torch.full((self.locs.shape[0],), float("inf"))

    def _positions(self, state_info):
        pos = self._scene_only(state_info, "centroid").float()
        if self._local_frame:
            if self._afw is None:
                self._afw = self._scene_only(state_info, "agent_from_world").float().clone()
            pos = _transform(pos, self._afw.to(pos.device))
        if self.agents is not None:
            pos = pos[torch.as_tensor(self.agents, device=pos.device)]
        return pos

    def add_step(self, state_info, all_scene_index):
        pos = self._positions(state_info)
        d = torch.linalg.norm(pos - self.locs.to(pos.device), dim=-1).cpu()
        self._min = torch.minimum(self._min, d)

    def get_episode_metrics(self):
        return self._for_scene(self._min.mean())


class GlobalTargetPosGuidance(TargetPosGuidance):
    """The same with targets in the world frame (guidance_metrics.py:220-247)."""
    _local_frame = False


class ConstraintGuidance(TargetPosGuidance):
    """Mean distance to the waypoint AT its target time; the first state seen is the one before the plan, so
    plan step t is entry t + 1 (guidance_metrics.py:249-296)."""

    def reset(self):
        if "locs" in self.params and "times" in self.params:
            locs, times = self.params["locs"], self.params["times"]
        elif "target_pos" in self.params and "target_time" in self.params:
            locs, times = self.params["target_pos"], self.params["target_time"]
        else:
            raise NotImplementedError("constraint metric needs locs/times or target_pos/target_time")
        self.locs = torch.as_tensor(np.array(locs), dtype=torch.float32)
        self.times = torch.as_tensor(np.array(times), dtype=torch.long) + 1
        self._afw = None
        self._per_step = []

    def add_step(self, state_info, all_scene_index):
        self._per_step.append(self._positions(state_info).cpu())

    def get_episode_metrics(self):
        allp = torch.stack(self._per_step, 1)                              # (agents, steps, 2)
        tgt = allp[torch.arange(allp.shape[0]), self.times]                # IndexError past the episode, as the reference
        return self._for_scene(torch.linalg.norm(tgt - self.locs, dim=-1).mean())


class GlobalConstraintGuidance(ConstraintGuidance):
    """guidance_metrics.py:298-338."""
    _local_frame = False


def _off_road(points_raster, drivable):
    """metrics.batch_detect_off_road (tbsim/utils/metrics.py:216-240): (n, k, 2) raster points, clamped to the
    image, looked up in the (n, H, W) drivable maps -> (n, k) 1 where off road."""
    n, H, W = drivable.shape
    x = points_raster[..., 0].clamp(0, W - 1).long()
    y = points_raster[..., 1].clamp(0, H - 1).long()
    d = torch.gather(drivable.reshape(n, -1).float(), 1, y * W + x)
    return 1.0 - d


class MapCollisionGuidance(GuidanceMetric):
    """Mean fraction of frames in which a corner of the agent's box lies off the drivable area of the raster the
    plan was made on (guidance_metrics.py:340-391; metrics.batch_detect_off_road_boxes)."""

    def reset(self):
        self._per_step = []

    def _raster_state(self, state_info):
        pos, yaw, ext, rfw, drv = self._select(state_info, "centroid", "yaw", "extent", "raster_from_world",
                                               "drivable_map")
        pos, yaw, ext, rfw = pos.float(), yaw.float().reshape(-1), ext.float()[..., :2], rfw.float()
        ctr = _transform(pos, rfw)
        # transform_yaw (geometry_utils): the heading of the rotated unit vector
        hx = rfw[:, 0, 0] * torch.cos(yaw) + rfw[:, 0, 1] * torch.sin(yaw)
        hy = rfw[:, 1, 0] * torch.cos(yaw) + rfw[:, 1, 1] * torch.sin(yaw)
        pix_per_m = torch.linalg.norm(rfw[:, :2, 0], dim=-1)               # get_raster_pix2m() of the reference
        return ctr, torch.atan2(hy, hx), ext * pix_per_m[:, None], drv

    def _off(self, state_info):
        ctr, yaw, ext, drv = self._raster_state(state_info)
        return (_off_road(_box_corners(ctr, yaw, ext), drv).sum(-1) > 0.5).float()

    def add_step(self, state_info, all_scene_index):
        self._per_step.append(self._off(state_info).cpu())

    def get_episode_metrics(self):
        allm = torch.stack(self._per_step, 1)
        return self._for_scene((allm.sum(1) / float(allm.shape[1])).mean())


class MapCollisionGuidanceDisk(MapCollisionGuidance):
    """The same with the agent as one disk sampled at 4 radii x 13 angles
    (guidance_metrics.py:393-443; metrics.batch_detect_off_road_disk)."""

    def _off(self, state_info):
        ctr, _, ext, drv = self._raster_state(state_info)
        th = torch.linspace(0, 2 * math.pi, 13, device=ctr.device)
        ring = torch.stack([torch.cos(th), torch.sin(th)], 1)              # (13, 2)
        rad = ext.amin(-1) / 2.0
        fr = torch.arange(1, 5, device=ctr.device, dtype=torch.float32) / 4.0      # linspace(0, r, 5)[1:]
        pts = ctr[:, None, None] + ring[None, None] * (rad[:, None] * fr[None])[..., None, None]
        return (_off_road(pts.reshape(ctr.shape[0], -1, 2), drv).sum(-1) > 0.5).float()


class SocialGroupGuidance(GuidanceMetric):
    """Mean | distance to the nearest other member - social_dist | over members and frames
    (guidance_metrics.py:445-481)."""

    def reset(self):
        self.social_dist = float(self.params["social_dist"])
        self._per_step = []

    def add_step(self, state_info, all_scene_index):
        pos = self._select(state_info, "centroid").float()
        d = torch.linalg.norm(pos[None] - pos[:, None], dim=-1)
        d = d.masked_fill(torch.eye(pos.shape[0], dtype=torch.bool, device=pos.device), float("inf"))
        self._per_step.append((d.amin(-1) - self.social_dist).abs().cpu())

    def get_episode_metrics(self):
        return self._for_scene(torch.stack(self._per_step, 1).mean())


# guidance_metrics.py:485-494
GUIDANCE_NAME_TO_METRICS = {
    "target_speed": {"target_speed": TargetSpeedGuidance},
    "agent_collision": {"agent_collision": AgentCollisionGuidance, "agent_collision_disk": AgentCollisionGuidanceDisk,
                        "social_dist": SocialDistanceGuidance},
    "map_collision": {"map_collision": MapCollisionGuidance, "map_collision_disk": MapCollisionGuidanceDisk},
    "target_pos_at_time": {"target_pos_at_time": ConstraintGuidance},
    "target_pos": {"target_pos": TargetPosGuidance},
    "global_target_pos_at_time": {"global_target_pos_at_time": GlobalConstraintGuidance},
    "global_target_pos": {"global_target_pos": GlobalTargetPosGuidance},
    "social_group": {"social_group": SocialGroupGuidance},
}


def guidance_metrics_from_config(guidance_config):
    """Metric objects that measure the success of each guidance (guidance_metrics.py:496-510): keys
    `guide_<metric>_s<scene>g<config index>`; guidance names without a metric of their own (min_speed) add none."""
    metrics = dict()
    num_scenes = len(guidance_config)
    for si in range(num_scenes):
        for ci, cur in enumerate(guidance_config[si]):
            for met_name, met_cls in GUIDANCE_NAME_TO_METRICS.get(cur["name"], {}).items():
                metrics["guide_%s_s%dg%d" % (met_name, si, ci)] = met_cls(si, cur["params"], cur["agents"], num_scenes)
    return metrics
