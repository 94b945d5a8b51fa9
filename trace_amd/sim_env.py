"""Device-resident closed-loop scene environment: the caller on either side of the sampler
(SURVEY.md section 8(f)-1).  It stands in for the reference's ``EnvUnifiedSimulation``
(/root/reference/tbsim/envs/env_trajdata.py:18-416) where datasets are absent: same gym-like
surface (``reset / get_observation / step / is_done / get_metrics / get_info /
reset_multi_episodes_metrics``, ``_current_scenes[i].dataset.desired_dt``), but the world state
(31-frame pose histories of all agents of all scenes) lives in HBM and the agent-centric observation
batch (input contract of SURVEY.md section 8(b), keys of trajdata_utils.py:238-270) is rebuilt
on the device after every step, so a replanning loop (BASELINE.json configs[4]) never leaves the GPU.

What follows the reference and where:
  * applying an ``Action``: env_trajdata.py:376-397 -- the plan is expressed in the agent frame of
    the observation it was made from; frame k moves the agent to ``pos[k] @ R + cur_xy`` with
    ``R = [[cos, sin], [-sin, cos]]`` and ``yaw = cur_yaw + yaws[k]``; ``num_steps_to_take`` frames
    are taken per call and the episode ends when ``frame_index + n >= num_simulation_steps`` (:401-404).
  * the observation: positions / yaws of the ego and neighbour histories in the CURRENT agent frame,
    speed = norm of the finite-difference velocity, unavailable frames NaN (the env parses with
    ``overwrite_nan=False``, env_trajdata.py:290), ``raster_from_agent`` from pixel size / ego centre
    (trajdata_utils.py:191-200), neighbours = same-scene agents within ``neighbor_radius`` (15 m for ORCA,
    orca_config.py:68) nearest first, ``drivable_map`` = layer 0 of the raster (trajdata_utils.py:97-99).
  * scenes are ORCA-style synthetic crowds; the "dataset future" used by the
    GT heuristics and ADE/FDE is each agent's constant-velocity continuation.

trajdata itself (collation, map crop/rotate) is not importable offline, so this file's parity is
pinned only by ``oracle/env_oracle.py`` (per-agent numpy restatement of the formulas cited above):
"parity unpinned" against trajdata's own rasteriser.
"""
import math

import numpy as np
import torch

from .config import ModelConfig
from .policy import RolloutAction


class _Dataset:
    def __init__(self, dt):
        self.desired_dt = dt


class SimScene:
    """Handle of one running scene: what ``compute_heuristic_guidance`` queries
    (scene_edit_utils.py:129-163,369-392 use ``sim_scene.cache`` / ``.agents`` / ``.dataset``)."""

    def __init__(self, env, slot, name):
        self._env, self._slot, self.scene_name = env, slot, name
        self.dataset = env.dataset
        lo, hi = env._scene_range(slot)
        self.agents = ["%s/agent_%03d" % (name, i) for i in range(hi - lo)]

    def get_agents_curr(self):
        """(A, 7) [x, y, vx, vy, ax, ay, heading] in the world frame at the initial frame
        (scene_edit_utils.py:151-163)."""
        e = self._env
        lo, hi = e._scene_range(self._slot)
        p = e._init_pose[lo:hi].cpu().numpy().astype(np.float64)
        v = e._init_speed[lo:hi].cpu().numpy().astype(np.float64)
        out = np.zeros((hi - lo, 7))
        out[:, :2] = p[:, :2]
        out[:, 2], out[:, 3] = v * np.cos(p[:, 2]), v * np.sin(p[:, 2])
        out[:, 6] = p[:, 2]
        return out

    def get_agents_future(self, fut_sec):
        """(A, T, 7) recorded future and (A, T) validity (scene_edit_utils.py:129-149)."""
        cur = self.get_agents_curr()
        T = int(round(fut_sec / self.dataset.desired_dt))
        k = np.arange(1, T + 1)[None, :, None] * self.dataset.desired_dt
        fut = np.repeat(cur[:, None], T, axis=1)
        fut[:, :, :2] = cur[:, None, :2] + cur[:, None, 2:4] * k
        return fut, np.ones((cur.shape[0], T), dtype=bool)


class SceneBatchEnv:
    """A pool of ``num_total_scenes`` synthetic scenes, ``num_scenes`` of them simulated at once."""

    def __init__(self, cfg: ModelConfig, agents_per_scene, device="cpu", seed=0, num_scenes=None,
                 num_simulation_steps=100, neighbor_radius=15.0, max_neighbors=16, area=30.0,
                 num_obstacles=3, dense=False, extent=(0.8, 0.8, 1.75), pixel_size=1.0 / 12.0,
                 ego_center=(-0.5, 0.0), with_image=True, metrics=None, name="orca_maps"):
        self.cfg, self.device = cfg, torch.device(device)
        self.name = name
        self._agents_per_scene_all = [int(a) for a in agents_per_scene]
        self._num_total_scenes = len(self._agents_per_scene_all)
        self._num_scenes = num_scenes if num_scenes is not None else self._num_total_scenes
        self._seed = seed
        self._npr = np.random.RandomState(seed)
        self.horizon = int(num_simulation_steps)
        self.neighbor_radius, self.max_neighbors = float(neighbor_radius), int(max_neighbors)
        self.area, self.num_obstacles, self.dense = float(area), int(num_obstacles), bool(dense)
        self._extent = tuple(extent)
        self.dataset = _Dataset(cfg.step_time)
        self.with_image = with_image
        C, H, W = cfg.image_shape
        res = 1.0 / pixel_size
        self._rfa = torch.tensor([[res, 0.0, ((1.0 + ego_center[0]) / 2.0) * W],
                                  [0.0, res, ((1.0 + ego_center[1]) / 2.0) * H],
                                  [0.0, 0.0, 1.0]], dtype=torch.float32, device=self.device)
        self._metrics = dict() if metrics is None else metrics
        self._current_scenes = None
        self._current_scene_indices = None
        self._done = False
        self._frame_index = 0
        self._cached_observation = None

    # ---- bookkeeping that mirrors the reference's properties (env_trajdata.py:62-116) ---------- #
    def update_random_seed(self, seed):
        self._npr = np.random.RandomState(seed)

    @property
    def num_instances(self):
        return self._num_scenes

    @property
    def total_num_scenes(self):
        return self._num_total_scenes

    @property
    def current_scene_index(self):
        return np.array(self._current_scene_indices).copy()

    @property
    def current_scene_names(self):
        return [s.scene_name for s in self._current_scenes]

    @property
    def current_num_agents(self):
        return int(self._scene_index.shape[0])

    @property
    def current_agent_scene_index(self):
        return self._scene_index.cpu().numpy()

    def is_done(self):
        return self._done

    def reset_multi_episodes_metrics(self):
        for v in self._metrics.values():
            v.multi_episode_reset()

    def _scene_range(self, slot):
        return int(self._scene_start[slot]), int(self._scene_start[slot + 1])

    # ---- scene generation --------------------------------------------------------------------- #
    def _make_scene(self, scene_id):
        """Seeded per scene id (independent of which other scenes run beside it, so results do not
        depend on how scenes are sharded over ranks)."""
        g = torch.Generator().manual_seed(self._seed * 1000003 + 7919 * int(scene_id) + 17)
        a = self._agents_per_scene_all[scene_id]
        if self.dense:
            side = int(math.ceil(math.sqrt(a)))
            ij = torch.stack(torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij"),
                             -1).reshape(-1, 2)[:a].float()
            pos = (ij - (side - 1) / 2.0) * 1.2 + (torch.rand(a, 2, generator=g) - 0.5) * 0.5
        else:
            pos = (torch.rand(a, 2, generator=g) - 0.5) * self.area
        yaw = (torch.rand(a, generator=g) * 2 - 1) * math.pi
        speed = 1.0 + torch.rand(a, generator=g)
        ctr = (torch.rand(self.num_obstacles, 2, generator=g) - 0.5) * self.area * (0.2 if self.dense else 1.0)
        half = 1.5 + 3.5 * torch.rand(self.num_obstacles, 2, generator=g)
        short = torch.rand(a, generator=g) < 0.25
        cut = torch.randint(1, self.cfg.hist_frames - 1, (a,), generator=g)
        first_avail = torch.where(short, cut, torch.zeros_like(cut))
        return pos, yaw, speed, torch.cat([ctr, half], -1), first_avail

    def reset(self, scene_indices=None, start_frame_index=None):
        """env_trajdata.py:139-212.  ``start_frame_index`` is accepted and ignored (synthetic scenes
        start at their generated frame)."""
        if scene_indices is None:
            scene_indices = self._npr.choice(np.arange(self._num_total_scenes), size=(self.num_instances,),
                                             replace=False)
        scene_indices = [int(s) for s in scene_indices]
        assert max(scene_indices) < self._num_total_scenes and min(scene_indices) >= 0
        self._num_scenes = len(scene_indices)
        self._current_scene_indices = scene_indices
        Th, dt = self.cfg.hist_frames, self.cfg.step_time
        poses, speeds, boxes, firsts, sidx = [], [], [], [], []
        for slot, sid in enumerate(scene_indices):
            pos, yaw, speed, box, first = self._make_scene(sid)
            poses.append(torch.cat([pos, yaw[:, None]], -1))
            speeds.append(speed)
            boxes.append(box)
            firsts.append(first)
            sidx.append(torch.full((pos.shape[0],), slot, dtype=torch.long))
        counts = [p.shape[0] for p in poses]
        self._scene_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        dev = self.device
        pose = torch.cat(poses).to(dev)
        speed = torch.cat(speeds).to(dev)
        self._init_pose, self._init_speed = pose.clone(), speed.clone()
        self._scene_index = torch.cat(sidx).to(dev)
        self._boxes = torch.stack(boxes).to(dev)                       # (S, K, 4) cx, cy, hx, hy
        B = pose.shape[0]
        self._extent_t = torch.tensor(self._extent, dtype=torch.float32, device=dev)[None].repeat(B, 1)
        # world-frame constant-velocity histories, frame Th-1 is "now"
        k = torch.arange(Th, dtype=torch.float32, device=dev) - (Th - 1)
        hv = torch.stack([torch.cos(pose[:, 2]), torch.sin(pose[:, 2])], -1)
        xy = pose[:, None, :2] + hv[:, None, :] * (speed[:, None, None] * dt * k[None, :, None])
        self._hist = torch.cat([xy, pose[:, None, 2:3].expand(B, Th, 1)], -1).contiguous()   # (B, Th, 3)
        self._hist_speed = speed[:, None].repeat(1, Th)
        first = torch.cat(firsts).to(dev)
        self._hist_avail = torch.arange(Th, device=dev)[None, :] >= first[:, None]
        # per-scene padded slots for the neighbour search: (S, Amax) agent id or -1
        amax = max(counts)
        slots = torch.full((len(counts), amax), -1, dtype=torch.long)
        for s, (lo, c) in enumerate(zip(self._scene_start[:-1], counts)):
            slots[s, :c] = torch.arange(lo, lo + c)
        self._slots = slots.to(dev)
        self._slot_of = (torch.arange(B) - torch.as_tensor(self._scene_start[:-1])[torch.cat(sidx)]).to(dev)
        self._current_scenes = [SimScene(self, s, "scene_%06d" % sid) for s, sid in enumerate(scene_indices)]
        self._frame_index, self._done, self._cached_observation = 0, False, None
        # episode accumulators for ADE / FDE against the recorded (constant-velocity) future
        self._err_sum = torch.zeros(B, device=dev)
        self._err_last = torch.zeros(B, device=dev)
        self._err_n = 0
        for v in self._metrics.values():
            v.reset()
        return [True] * len(scene_indices)

    # ---- observation builder ------------------------------------------------------------------ #
    def _rasterize(self, pose):
        """image (B, C, H, W) float and drivable (B, H, W) bool of every agent's frame.  On a GPU this is
        the HIP kernel ``trace_env_rasterize`` (csrc/steps.hip; raises if the library is missing);
        ``_rasterize_torch`` is the host-logic restatement the CPU tests run."""
        C, H, W = self.cfg.image_shape
        B = pose.shape[0]
        if self.device.type != "cuda":
            drv = self._rasterize_torch(pose)
            img = torch.zeros((B, C, H, W), dtype=torch.float32, device=self.device)
            img[:, 0] = drv
            if C > 1:
                img[:, 1] = ~drv
            return img, drv
        import ctypes
        from . import _cabi
        L = _cabi.lib()
        img = torch.empty((B, C, H, W), dtype=torch.float32, device=self.device)
        drv = torch.empty((B, H, W), dtype=torch.uint8, device=self.device)
        pose_c = pose.contiguous()
        soa = self._scene_index.to(torch.int32).contiguous()
        boxes = self._boxes.contiguous()
        rfa = (ctypes.c_float * 9)(*[float(v) for v in self._rfa.flatten().tolist()])
        st = torch.cuda.current_stream(self.device).cuda_stream
        for lo in range(0, B, 65535):
            hi = min(B, lo + 65535)
            _cabi.check(L.trace_env_rasterize(pose_c[lo:hi].data_ptr(), soa[lo:hi].data_ptr(), boxes.data_ptr(),
                                              boxes.shape[1], ctypes.byref(rfa), img[lo:hi].data_ptr(), C,
                                              drv[lo:hi].data_ptr(), hi - lo, H, W, st), "trace_env_rasterize")
        return img, drv.view(torch.bool)

    def _rasterize_torch(self, pose):
        """(B, H, W) bool drivable layer in each agent's frame: pixel centres -> agent -> world, then
        inside tests against the scene's boxes.  Chunked so the temporaries stay below ~1 GB."""
        C, H, W = self.cfg.image_shape
        dev = self.device
        rfa = self._rfa
        py, px = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev),
                                torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
        ax = ((px - rfa[0, 2]) / rfa[0, 0]).reshape(1, -1)
        ay = ((py - rfa[1, 2]) / rfa[1, 1]).reshape(1, -1)
        B = pose.shape[0]
        out = torch.empty(B, H * W, dtype=torch.bool, device=dev)
        chunk = max(1, (1 << 26) // (H * W))
        for lo in range(0, B, chunk):
            p = pose[lo:lo + chunk]
            c, s = torch.cos(p[:, 2:3]), torch.sin(p[:, 2:3])
            wx = c * ax - s * ay + p[:, 0:1]
            wy = s * ax + c * ay + p[:, 1:2]
            box = self._boxes[self._scene_index[lo:lo + chunk]]        # (b, K, 4)
            drv = torch.ones_like(wx, dtype=torch.bool)
            for q in range(box.shape[1]):
                inside = ((wx - box[:, q, 0:1]).abs() < box[:, q, 2:3]) & ((wy - box[:, q, 1:2]).abs() < box[:, q, 3:4])
                drv &= ~inside
            out[lo:lo + chunk] = drv
        return out.reshape(B, H, W)

    def get_observation(self):
        if self._cached_observation is not None:
            return self._cached_observation
        dev = self.device
        hist, B, Th = self._hist, self._hist.shape[0], self.cfg.hist_frames
        pose = hist[:, -1]                                             # (B, 3) current world pose
        c, s = torch.cos(pose[:, 2]), torch.sin(pose[:, 2])
        zero, one = torch.zeros_like(c), torch.ones_like(c)
        wfa = torch.stack([torch.stack([c, -s, pose[:, 0]], -1), torch.stack([s, c, pose[:, 1]], -1),
                           torch.stack([zero, zero, one], -1)], -2)
        tx = -(c * pose[:, 0] + s * pose[:, 1])
        ty = -(-s * pose[:, 0] + c * pose[:, 1])
        afw = torch.stack([torch.stack([c, s, tx], -1), torch.stack([-s, c, ty], -1),
                           torch.stack([zero, zero, one], -1)], -2)
        nan = float("nan")

        def to_local(h, cc, ss, p):
            """world (…, Th, 3) histories into the frame of pose p: broadcast over leading dims"""
            dx, dy = h[..., 0] - p[..., 0:1], h[..., 1] - p[..., 1:2]
            lx = cc * dx + ss * dy
            ly = -ss * dx + cc * dy
            dyaw = h[..., 2] - p[..., 2:3]
            return torch.stack([lx, ly], -1), torch.atan2(torch.sin(dyaw), torch.cos(dyaw)).unsqueeze(-1)

        hp, hy = to_local(hist, c[:, None], s[:, None], pose)
        av = self._hist_avail
        hist_pos = torch.where(av[..., None], hp, torch.full_like(hp, nan))
        hist_yaw = torch.where(av[..., None], hy, torch.full_like(hy, nan))
        hist_speed = torch.where(av, self._hist_speed, torch.full_like(self._hist_speed, nan))

        # neighbours: per scene all-pairs distances on the padded slots, nearest first
        slots = self._slots                                            # (S, A)
        S, A = slots.shape
        valid = slots >= 0
        sp = pose[slots.clamp(min=0)][..., :2]                         # (S, A, 2)
        d = torch.cdist(sp, sp)                                        # (S, A, A)
        ok = valid[:, :, None] & valid[:, None, :] & ~torch.eye(A, dtype=torch.bool, device=dev)[None]
        ok &= d < self.neighbor_radius
        d = torch.where(ok, d, torch.full_like(d, float("inf")))
        Nn = max(1, min(self.max_neighbors, A - 1))
        if A > 1:
            dn, jn = torch.topk(d, Nn, dim=-1, largest=False, sorted=True)     # (S, A, Nn)
        else:
            dn = torch.full((S, A, Nn), float("inf"), device=dev)
            jn = torch.zeros((S, A, Nn), dtype=torch.long, device=dev)
        sid, slot = self._scene_index, self._slot_of
        nb_ok = torch.isfinite(dn[sid, slot])                          # (B, Nn)
        nb_id = slots[sid[:, None], jn[sid, slot]].clamp(min=0)        # (B, Nn)
        nh = hist[nb_id]                                               # (B, Nn, Th, 3)
        np_, ny = to_local(nh, c[:, None, None], s[:, None, None], pose[:, None, :])
        nav = self._hist_avail[nb_id] & nb_ok[..., None]
        n_pos = torch.where(nav[..., None], np_, torch.full_like(np_, nan))
        n_yaw = torch.where(nav[..., None], ny, torch.full_like(ny, nan))
        n_speed = torch.where(nav, self._hist_speed[nb_id], torch.full((), nan, device=dev))
        n_ext = torch.where(nb_ok[..., None], self._extent_t[nb_id], torch.zeros((), device=dev))

        obs = {
            "history_positions": hist_pos, "history_yaws": hist_yaw, "history_speeds": hist_speed,
            "history_availabilities": av.clone(), "extent": self._extent_t.clone(),
            "curr_speed": self._hist_speed[:, -1].clone(),
            "all_other_agents_history_positions": n_pos, "all_other_agents_history_yaws": n_yaw,
            "all_other_agents_history_speeds": n_speed,
            "all_other_agents_history_availabilities": nav, "all_other_agents_history_availability": nav,
            "all_other_agents_extents": n_ext,
            "raster_from_agent": self._rfa[None].repeat(B, 1, 1), "world_from_agent": wfa,
            "agent_from_world": afw, "raster_from_world": self._rfa[None] @ afw,
            "scene_index": torch.as_tensor(self.current_scene_index, device=dev)[self._scene_index],
            "track_id": torch.arange(B, device=dev),
            "centroid": pose[:, :2].clone(), "yaw": pose[:, 2].clone(),
            "curr_agent_state": pose.clone(),
        }
        # recorded future ("ground truth": the constant-velocity continuation of the initial state) in
        # the CURRENT agent frame: target_positions / target_yaws / target_availabilities of the parsed
        # batch (trajdata_utils.py:241-243), consumed by guide_with_gt and the GT replay policy
        T = self.cfg.horizon
        kk = (self._frame_index + 1 + torch.arange(T, device=dev, dtype=torch.float32)) * self.cfg.step_time
        hv0 = torch.stack([torch.cos(self._init_pose[:, 2]), torch.sin(self._init_pose[:, 2])], -1)
        fut_w = self._init_pose[:, None, :2] + hv0[:, None] * (self._init_speed[:, None, None] * kk[None, :, None])
        fut = torch.cat([fut_w, self._init_pose[:, None, 2:3].expand(B, T, 1)], -1)
        fp, fy = to_local(fut, c[:, None], s[:, None], pose)
        obs["target_positions"], obs["target_yaws"] = fp, fy
        obs["target_availabilities"] = torch.ones(B, T, dtype=torch.bool, device=dev)
        if self.with_image:
            obs["image"], obs["drivable_map"] = self._rasterize(pose)
        self._cached_observation = dict(agents=obs)
        return self._cached_observation

    # ---- stepping ------------------------------------------------------------------------------ #
    def step(self, actions: RolloutAction, num_steps_to_take=1, render=False):
        """env_trajdata.py:336-416.  The action stays on the device (the reference converts it to
        numpy first, :414)."""
        if self.is_done():
            raise RuntimeError("Cannot step in a finished episode")
        obs = self.get_observation()["agents"]
        pos = torch.as_tensor(actions.agents.positions, device=self.device, dtype=torch.float32)
        yaws = torch.as_tensor(actions.agents.yaws, device=self.device, dtype=torch.float32)
        n = min(int(num_steps_to_take), pos.shape[1])
        cur = obs["curr_agent_state"]
        c, s = torch.cos(cur[:, 2:3]), torch.sin(cur[:, 2:3])
        # next_xy = a @ [[c, s], [-s, c]] + cur_xy   (row vector times matrix, :383-391)
        wx = pos[:, :n, 0] * c - pos[:, :n, 1] * s + cur[:, 0:1]
        wy = pos[:, :n, 0] * s + pos[:, :n, 1] * c + cur[:, 1:2]
        wyaw = cur[:, 2:3] + yaws[:, :n, 0]
        new = torch.stack([wx, wy, wyaw], -1)                          # (B, n, 3)
        prev = torch.cat([self._hist[:, -1:, :2], new[:, :-1, :2]], 1)
        spd = torch.linalg.norm(new[..., :2] - prev, dim=-1) / self.cfg.step_time
        # Metrics see the state BEFORE each action is applied, with the map of the observation the plan was made
        # from (env_trajdata.py:345-358): frame 0 is the current state, frame k the state after k actions.
        if self._metrics:
            pre = torch.cat([cur[:, None, :], new[:, :n - 1]], 1)          # (B, n, 3)
            pre_spd = torch.cat([self._hist_speed[:, -1:], spd[:, :n - 1]], 1)
            for k in range(n):
                pk = pre[:, k]
                ck, sk = torch.cos(pk[:, 2]), torch.sin(pk[:, 2])
                afw = torch.zeros(pk.shape[0], 3, 3, device=self.device)
                afw[:, 0, 0], afw[:, 0, 1], afw[:, 1, 0], afw[:, 1, 1], afw[:, 2, 2] = ck, sk, -sk, ck, 1.0
                afw[:, 0, 2] = -(ck * pk[:, 0] + sk * pk[:, 1])
                afw[:, 1, 2] = -(-sk * pk[:, 0] + ck * pk[:, 1])
                step_obs = {"centroid": pk[:, :2], "yaw": pk[:, 2], "extent": self._extent_t,
                            "curr_speed": pre_spd[:, k], "agent_from_world": afw,
                            "scene_index": obs["scene_index"], "boxes": self._boxes[self._scene_index],
                            "raster_from_world": obs["raster_from_world"]}
                if "drivable_map" in obs:
                    step_obs["drivable_map"], step_obs["image"] = obs["drivable_map"], obs["image"]
                for m in self._metrics.values():
                    m.add_step(step_obs, self.current_scene_index)
        # ADE / FDE against the recorded future
        dt = self.cfg.step_time
        kk = (self._frame_index + 1 + torch.arange(n, device=self.device, dtype=torch.float32)) * dt
        hv = torch.stack([torch.cos(self._init_pose[:, 2]), torch.sin(self._init_pose[:, 2])], -1)
        gt = self._init_pose[:, None, :2] + hv[:, None] * (self._init_speed[:, None, None] * kk[None, :, None])
        err = torch.linalg.norm(new[..., :2] - gt, dim=-1)
        self._err_sum += err.sum(1)
        self._err_last = err[:, -1]
        self._err_n += n
        Th = self.cfg.hist_frames
        self._hist = torch.cat([self._hist, new], 1)[:, -Th:].contiguous()
        self._hist_speed = torch.cat([self._hist_speed, spd], 1)[:, -Th:].contiguous()
        self._hist_avail = torch.cat([self._hist_avail, torch.ones_like(spd, dtype=torch.bool)], 1)[:, -Th:].contiguous()
        self._cached_observation = None
        if n < num_steps_to_take or self._frame_index + num_steps_to_take >= self.horizon:
            self._done = True
            self._frame_index += n if n < num_steps_to_take else 0
        else:
            self._frame_index += num_steps_to_take
        return []

    # ---- results -------------------------------------------------------------------------------- #
    def _per_scene_mean(self, v):
        S = self._num_scenes
        tot = torch.zeros(S, device=self.device).index_add_(0, self._scene_index, v)
        cnt = torch.zeros(S, device=self.device).index_add_(0, self._scene_index, torch.ones_like(v))
        return (tot / cnt.clamp(min=1)).cpu().numpy()

    def get_metrics(self):
        """env_trajdata.py:246-272: one value per scene for every metric."""
        metrics = {"ade": self._per_scene_mean(self._err_sum / max(self._err_n, 1)),
                   "fde": self._per_scene_mean(self._err_last)}
        for name, met in self._metrics.items():
            vals = met.get_episode_metrics()
            if isinstance(vals, dict):
                for k, v in vals.items():
                    metrics[name + "_" + k] = v
            else:
                metrics[name] = vals
        for k in metrics:
            assert metrics[k].shape == (self.num_instances,), k
        return metrics

    def get_info(self):
        return dict(scene_index=self.current_scene_names)


class CollisionRate:
    """Per-scene fraction of agents that were ever closer to another agent of their scene than the
    sum of their radii (disk per agent, width / 2).  Accumulated on the device per simulated frame."""

    def __init__(self):
        self._hit = None
        self._episodes = []

    def reset(self):
        self._hit = None

    def multi_episode_reset(self):
        self._episodes = []

    def add_step(self, obs, scene_index):
        p, sid, r = obs["centroid"], obs["scene_index"], obs["extent"][:, 1] / 2.0
        B, dev = p.shape[0], p.device
        _, inv, counts = torch.unique_consecutive(sid, return_inverse=True, return_counts=True)
        self._inv = inv
        # padded per-scene layout (S, Amax): one batched all-pairs test instead of a loop over scenes
        S, A = counts.shape[0], int(counts.max())
        start = torch.cumsum(counts, 0) - counts
        slot = torch.arange(B, device=dev) - start[inv]
        pp = torch.full((S, A, 2), float("inf"), device=dev)
        rr = torch.zeros(S, A, device=dev)
        pp[inv, slot] = p
        rr[inv, slot] = r
        d = torch.linalg.norm(pp[:, :, None] - pp[:, None, :], dim=-1) - (rr[:, :, None] + rr[:, None, :])
        d = torch.where(torch.eye(A, dtype=torch.bool, device=dev)[None], torch.full_like(d, float("inf")), d)
        hit = (torch.nan_to_num(d, nan=float("inf")) < 0).any(-1)[inv, slot]
        self._hit = hit if self._hit is None else (self._hit | hit)

    def get_episode_metrics(self):
        inv, hit = self._inv, self._hit.float()
        S = int(inv.max()) + 1
        tot = torch.zeros(S, device=hit.device).index_add_(0, inv, hit)
        cnt = torch.zeros(S, device=hit.device).index_add_(0, inv, torch.ones_like(hit))
        return (tot / cnt).cpu().numpy()


class OffRoadRate:
    """Per-scene mean fraction of simulated frames an agent's centre spent inside an obstacle."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._sum, self._n = None, 0

    def multi_episode_reset(self):
        self.reset()

    def add_step(self, obs, scene_index):
        p, box = obs["centroid"], obs["boxes"]
        inside = ((p[:, None, 0] - box[..., 0]).abs() < box[..., 2]) & ((p[:, None, 1] - box[..., 1]).abs() < box[..., 3])
        off = inside.any(1).float()
        _, self._inv = torch.unique_consecutive(obs["scene_index"], return_inverse=True)
        self._sum = off if self._sum is None else self._sum + off
        self._n += 1

    def get_episode_metrics(self):
        inv, v = self._inv, self._sum / max(self._n, 1)
        S = int(inv.max()) + 1
        tot = torch.zeros(S, device=v.device).index_add_(0, inv, v)
        cnt = torch.zeros(S, device=v.device).index_add_(0, inv, torch.ones_like(v))
        return (tot / cnt).cpu().numpy()
