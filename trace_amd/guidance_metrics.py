"""Per-guidance success metrics of a closed-loop rollout: the `guide_<metric>_s<scene>g<config>` entries the
reference adds to the environment's metrics while a guidance configuration is active and reports in `stats.json`.

Behaviour restated from /root/reference/tbsim/utils/guidance_metrics.py (class by class, lines cited below) and wired
in as /root/reference/tbsim/utils/scene_edit_utils.py:47-60 does (`guidance_metrics_from_config` -> `env._metrics`).
Every metric handles ONLY the scene (local index inside the batch) and the agents of its guidance config, is fed the
state BEFORE each simulated action (env_trajdata.py:345-358: the first state it sees is the one the plan was made
from), and returns one value per scene of the batch, NaN everywhere but its own scene.

Differences by construction:
* `add_step` only RECORDS the frame -- numpy views of the metric's scene rows; the per-agent arrays of a frame are
  brought to the host once per frame and shared by all metric objects (the reference converts the whole observation
  to numpy per step, env_trajdata.py:352) -- and `get_episode_metrics` evaluates all frames at once.  A closed loop of
  128 scenes carries ~800 metric objects for 100 frames; evaluated eagerly on device tensors they cost a third of the
  episode (79 s against 59 s), recorded and evaluated at the end they cost nothing measurable.
* the oriented-box collision test is a separating-axis test (the reference asks shapely whether the two polygons
  intersect -- the same predicate up to boundary contact);
* the drivable map may stay a device tensor: only the pixel indices go to the device and the looked-up values back;
* rendering / data-frame hooks of `EnvMetrics` are not part of this path.
"""
import math

import numpy as np
import torch

_PER_AGENT = ("centroid", "yaw", "extent", "curr_speed", "agent_from_world", "raster_from_world")


def _np(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _host(state_info):
    """Host copies of a frame's per-agent arrays + each local scene's row range, made once per frame dictionary and
    shared by every metric object (cached in the dictionary under a private key)."""
    h = state_info.get("_guidance_metrics_host")
    if h is None:
        h = {k: _np(state_info[k]) for k in _PER_AGENT if k in state_info}
        sid = _np(state_info["scene_index"])
        # local index = rank of each run of equal ids (guidance_metrics.py:50: unique_consecutive ... inverse)
        change = np.flatnonzero(sid[1:] != sid[:-1]) + 1
        starts = np.concatenate([[0], change])
        stops = np.concatenate([change, [sid.shape[0]]])
        h["_ranges"] = list(zip(starts.tolist(), stops.tolist()))
        state_info["_guidance_metrics_host"] = h
    return h


class GuidanceMetric:
    """guidance_metrics.py:22-38."""
    _keys = ()

    def __init__(self, scene_idx, params, agents, num_scenes):
        self.scene_idx = int(scene_idx)
        self.params = params
        self.agents = None if agents is None else np.asarray([int(a) for a in agents], np.int64)
        self.num_scenes = int(num_scenes)
        self.reset()

    # the environment's metric protocol (env_metrics.EnvMetrics)
    def reset(self):
        self._frames = []

    def multi_episode_reset(self):
        self.reset()

    def add_step(self, state_info, all_scene_index):
        h = _host(state_info)
        lo, hi = h["_ranges"][self.scene_idx]
        self._frames.append(self._record(state_info, h, lo, hi))

    def _record(self, state_info, h, lo, hi):
        return tuple(h[k][lo:hi] for k in self._keys)            # views: nothing is copied per metric object

    def _stack(self, i, own_agents=True):
        """(frames, agents, ...) of recorded item i, restricted to the guidance's agents when asked."""
        v = np.stack([f[i] for f in self._frames], 0)
        if own_agents and self.agents is not None:
            v = v[:, self.agents]
        return v

    def _for_scene(self, value):
        met = np.ones(self.num_scenes) * np.nan
        met[self.scene_idx] = float(value)
        return met


class TargetSpeedGuidance(GuidanceMetric):
    """Mean |speed - target_speed| of the guided agents over the episode (guidance_metrics.py:40-69)."""
    _keys = ("curr_speed",)

    def get_episode_metrics(self):
        speed = self._stack(0).astype(np.float64)                            # (F, n)
        dev = np.abs(speed - np.asarray(self.params["target_speed"], np.float64)[None])
        return self._for_scene(dev.mean(1).mean())                           # mean over frames of per-frame means


def _box_corners(pos, yaw, extent):
    """(..., 4, 2) corners of oriented boxes: extent = (length, width) along / across the heading
    (geometry_utils.get_box_world_coords: corner offsets times the rotation, plus the centre)."""
    c, s = np.cos(yaw)[..., None], np.sin(yaw)[..., None]
    sx = np.array([-1.0, -1.0, 1.0, 1.0], pos.dtype) * 0.5
    sy = np.array([-1.0, 1.0, 1.0, -1.0], pos.dtype) * 0.5
    dx, dy = sx * extent[..., 0:1], sy * extent[..., 1:2]
    x = dx * c - dy * s + pos[..., 0:1]
    y = dx * s + dy * c + pos[..., 1:2]
    return np.stack([x, y], -1)


def boxes_intersect(pos, yaw, extent):
    """(..., n, n) bool: do the oriented boxes of agents i and j overlap (separating-axis test on the four edge
    normals); leading dimensions (frames) broadcast."""
    pos, yaw, extent = (_np(pos).astype(np.float64), _np(yaw).astype(np.float64), _np(extent).astype(np.float64))
    n = pos.shape[-2]
    cor = _box_corners(pos, yaw, extent)                                     # (..., n, 4, 2)
    ax = np.stack([np.stack([np.cos(yaw), np.sin(yaw)], -1),
                   np.stack([-np.sin(yaw), np.cos(yaw)], -1)], -2)           # (..., n, 2 axes, 2)
    proj = np.einsum("...aud,...bkd->...abuk", ax, cor)                      # axes of box a, corners of box b
    own = np.einsum("...aud,...akd->...auk", ax, cor)                        # axes of box a, its own corners
    omin, omax = own.min(-1)[..., :, None, :], own.max(-1)[..., :, None, :]  # (..., a, 1, u)
    pmin, pmax = proj.min(-1), proj.max(-1)                                  # (..., a, b, u)
    sep = ((omax < pmin) | (pmax < omin)).any(-1)                            # an axis of a separates a and b
    hit = ~(sep | np.swapaxes(sep, -1, -2))
    hit &= ~np.eye(n, dtype=bool)
    return hit


class AgentCollisionGuidance(GuidanceMetric):
    """Fraction of the guided agents whose box ever overlapped another agent of the scene
    (guidance_metrics.py:71-122)."""
    _keys = ("centroid", "yaw", "extent")

    def _coll(self):
        pos, yaw, ext = self._stack(0, False), self._stack(1, False), self._stack(2, False)[..., :2]
        return boxes_intersect(pos, yaw.reshape(yaw.shape[0], -1), ext).any(-1)       # (F, n)

    def get_episode_metrics(self):
        coll = self._coll().astype(np.float64)
        if self.agents is not None:
            coll = coll[:, self.agents]
        return self._for_scene(coll.max(0).mean())


class AgentCollisionGuidanceDisk(AgentCollisionGuidance):
    """The same with one disk per agent (radius = min extent / 2), optionally widened by the guidance's
    buffer distance (guidance_metrics.py:124-182)."""

    def __init__(self, scene_idx, params, agents, num_scenes, use_buffer_dist=False):
        self.use_buffer_dist = use_buffer_dist
        super().__init__(scene_idx, params, agents, num_scenes)

    def _coll(self):
        pos = self._stack(0, False).astype(np.float64)
        rad = self._stack(2, False)[..., :2].astype(np.float64).min(-1) / 2.0
        d = np.linalg.norm(pos[:, :, None] - pos[:, None], axis=-1)
        allowed = rad[:, :, None] + rad[:, None]
        if self.use_buffer_dist:
            allowed = allowed + float(self.params["buffer_dist"])
        hit = (d < allowed) & ~np.eye(pos.shape[1], dtype=bool)
        return hit.any(-1)


class SocialDistanceGuidance(AgentCollisionGuidanceDisk):
    def __init__(self, scene_idx, params, agents, num_scenes):
        super().__init__(scene_idx, params, agents, num_scenes, use_buffer_dist=True)


def _transform(points, mat):
    """(..., n, 2) points through (n, 3, 3) homogeneous transforms (geometry_utils.batch_nd_transform_points)."""
    return np.einsum("nij,...nj->...ni", mat[:, :2, :2], points) + mat[:, :2, 2]


class TargetPosGuidance(GuidanceMetric):
    """Mean over the guided agents of the smallest distance to their target, measured in the agent frame of
    the FIRST state seen (the planning frame the target was given in) (guidance_metrics.py:184-218)."""
    _local_frame = True
    _keys = ("centroid", "agent_from_world")

    def _positions(self):
        pos = self._stack(0, False).astype(np.float64)                       # (F, n, 2)
        if self._local_frame:
            pos = _transform(pos, self._frames[0][1].astype(np.float64))
        if self.agents is not None:
            pos = pos[:, self.agents]
        return pos

    def get_episode_metrics(self):
        locs = np.asarray(self.params["target_pos"], np.float64)
        d = np.linalg.norm(self._positions() - locs[None], axis=-1)          # (F, agents)
        return self._for_scene(d.min(0).mean())


class GlobalTargetPosGuidance(TargetPosGuidance):
    """The same with targets in the world frame (guidance_metrics.py:220-247)."""
    _local_frame = False


class ConstraintGuidance(TargetPosGuidance):
    """Mean distance to the waypoint AT its target time; the first state seen is the one before the plan, so
    plan step t is entry t + 1 (guidance_metrics.py:249-296)."""

    def reset(self):
        super().reset()
        if "locs" in self.params and "times" in self.params:
            locs, times = self.params["locs"], self.params["times"]
        elif "target_pos" in self.params and "target_time" in self.params:
            locs, times = self.params["target_pos"], self.params["target_time"]
        else:
            raise NotImplementedError("constraint metric needs locs/times or target_pos/target_time")
        self.locs = np.asarray(locs, np.float64)
        self.times = np.asarray(times, np.int64) + 1

    def get_episode_metrics(self):
        allp = self._positions()                                             # (F, agents, 2)
        tgt = allp[self.times, np.arange(allp.shape[1])]                     # IndexError past the episode, as the reference
        return self._for_scene(np.linalg.norm(tgt - self.locs, axis=-1).mean())


class GlobalConstraintGuidance(ConstraintGuidance):
    """guidance_metrics.py:298-338."""
    _local_frame = False


class MapCollisionGuidance(GuidanceMetric):
    """Mean fraction of frames in which a corner of the agent's box lies off the drivable area of the raster the
    plan was made on (guidance_metrics.py:340-391; metrics.batch_detect_off_road_boxes).  Raster coordinates in
    fp32 with the reference's operation order (the pixel index is a truncation)."""

    def reset(self):
        super().reset()
        self._raster = self._raster_rows = None

    def _record(self, state_info, h, lo, hi):
        rows = np.arange(lo, hi) if self.agents is None else lo + self.agents
        # Only the metric's own rows of a raster batch are kept (flattened, on the raster's device), sliced once per
        # raster object: a long closed loop would otherwise pin one full (agents, H, W) batch per replan.  The latest
        # batch object itself stays referenced so that the next frame can tell whether the raster changed.
        drv = state_info["drivable_map"]
        if self._raster is not drv:
            self._raster = drv
            if torch.is_tensor(drv):
                m = drv.reshape(drv.shape[0], -1)[torch.as_tensor(rows, device=drv.device)].clone()
            else:
                m = np.asarray(drv).reshape(np.asarray(drv).shape[0], -1)[rows].copy()
            self._raster_rows = (m, int(drv.shape[-2]), int(drv.shape[-1]))
        return (h["centroid"][rows], h["yaw"][rows], h["extent"][rows], h["raster_from_world"][rows],
                self._raster_rows, rows)

    def _points(self, ctr, yaw, ext):
        return _box_corners(ctr, yaw, ext)                                   # (F, n, 4, 2)

    def get_episode_metrics(self):
        off = []
        i = 0
        while i < len(self._frames):                                         # runs of frames that share a raster
            j = i
            while j < len(self._frames) and self._frames[j][4] is self._frames[i][4]:
                j += 1
            fr = self._frames[i:j]
            pos = np.stack([f[0] for f in fr]).astype(np.float32)            # (F, n, 2)
            yaw = np.stack([f[1] for f in fr]).astype(np.float32).reshape(len(fr), -1)
            ext = np.stack([f[2] for f in fr]).astype(np.float32)[..., :2]
            rfw = np.stack([f[3] for f in fr]).astype(np.float32)            # (F, n, 3, 3)
            ctr = (np.einsum("fnij,fnj->fni", rfw[..., :2, :2], pos) + rfw[..., :2, 2]).astype(np.float32)
            # transform_yaw (geometry_utils): the heading of the rotated unit vector
            hx = rfw[..., 0, 0] * np.cos(yaw) + rfw[..., 0, 1] * np.sin(yaw)
            hy = rfw[..., 1, 0] * np.cos(yaw) + rfw[..., 1, 1] * np.sin(yaw)
            pix_per_m = np.linalg.norm(rfw[..., :2, 0], axis=-1).astype(np.float32)  # get_raster_pix2m() of the reference
            pts = self._points(ctr, np.arctan2(hy, hx).astype(np.float32), ext * pix_per_m[..., None])
            (m, H, W), rows = fr[0][4], fr[0][5]
            x = np.clip(pts[..., 0], 0, W - 1).astype(np.int64)              # metrics.batch_detect_off_road
            y = np.clip(pts[..., 1], 0, H - 1).astype(np.int64)
            idx = np.moveaxis(y * W + x, 0, 1).reshape(len(rows), -1)        # (n, F * k)
            if torch.is_tensor(m):
                val = torch.gather(m.float(), 1, torch.as_tensor(idx, device=m.device)).cpu().numpy()
            else:
                val = np.take_along_axis(m.astype(np.float32), idx, 1)
            offp = (1.0 - val).reshape(len(rows), len(fr), -1)               # (n, F, k): 1 where off road
            off.append((offp.sum(-1) > 0.5).astype(np.float64))
            i = j
        allm = np.concatenate(off, 1)                                        # (n, frames)
        return self._for_scene((allm.sum(1) / float(allm.shape[1])).mean())


class MapCollisionGuidanceDisk(MapCollisionGuidance):
    """The same with the agent as one disk sampled at 4 radii x 13 angles
    (guidance_metrics.py:393-443; metrics.batch_detect_off_road_disk)."""

    def _points(self, ctr, yaw, ext):
        th = torch.linspace(0, 2 * math.pi, 13).numpy()                      # the reference's own angles, fp32
        ring = np.stack([np.cos(th), np.sin(th)], 1).astype(np.float32)      # (13, 2)
        rad = ext.min(-1) / np.float32(2.0)                                  # (F, n)
        fr = np.arange(1, 5, dtype=np.float32) / np.float32(4.0)             # linspace(0, r, 5)[1:]
        pts = ctr[:, :, None, None] + ring[None, None, None] * (rad[..., None] * fr)[..., None, None]
        return pts.reshape(ctr.shape[0], ctr.shape[1], -1, 2)


class SocialGroupGuidance(GuidanceMetric):
    """Mean | distance to the nearest other member - social_dist | over members and frames
    (guidance_metrics.py:445-481)."""
    _keys = ("centroid",)

    def get_episode_metrics(self):
        pos = self._stack(0).astype(np.float64)                              # (F, members, 2)
        d = np.linalg.norm(pos[:, None] - pos[:, :, None], axis=-1)
        d[:, np.arange(d.shape[1]), np.arange(d.shape[1])] = np.inf
        return self._for_scene(np.abs(d.min(-1) - float(self.params["social_dist"])).mean())


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
