"""TEST INFRASTRUCTURE ONLY -- per-agent numpy restatement of the closed-loop environment arithmetic
either side of the sampler.  Imported only by tests/ (never by the product path).

  * ``apply_action``  follows /root/reference/tbsim/envs/env_trajdata.py:376-397 literally (one agent and
    one frame at a time, the 2x2 matrix product written as in the source).
  * ``agent_frame``   follows the agent-centric convention of the parsed batch
    (/root/reference/tbsim/utils/trajdata_utils.py:155-273: positions / headings relative to the current
    pose, ``world_from_agent = inverse(agents_from_world_tf)``, raster_from_agent :196-200) and
    /root/reference/tbsim/utils/scene_edit_utils.py:165-179 for the transform itself.

trajdata's own collation is not importable offline (kornia / cv2 absent), so for the observation
this oracle is "parity unpinned": it restates the published convention, it was not checked against
trajdata outputs.
"""
import numpy as np


def apply_action(curr_agent_state, positions, yaws, num_steps):
    """curr_agent_state (B, 3) world x, y, yaw at planning time; positions (B, T, 2), yaws (B, T, 1)
    in that agent frame.  Returns (B, num_steps, 3) world states."""
    B = positions.shape[0]
    out = np.full((B, num_steps, 3), np.nan, dtype=positions.dtype)
    for k in range(num_steps):
        for idx in range(B):
            curr_yaw = curr_agent_state[idx, -1]
            curr_pos = curr_agent_state[idx, :2]
            world_from_agent = np.array([[np.cos(curr_yaw), np.sin(curr_yaw)],
                                         [-np.sin(curr_yaw), np.cos(curr_yaw)]])
            if not np.any(np.isnan(positions[idx, k])):
                out[idx, k, :2] = positions[idx, k] @ world_from_agent + curr_pos
                out[idx, k, 2] = curr_yaw + yaws[idx, k, 0]
    return out


def agent_frame(hist_world, pose):
    """hist_world (Th, 3) world x, y, yaw; pose (3,) current world pose of the observing agent.
    Returns local positions (Th, 2) and wrapped relative headings (Th,)."""
    c, s = np.cos(pose[2]), np.sin(pose[2])
    world_from_agent = np.array([[c, -s, pose[0]], [s, c, pose[1]], [0.0, 0.0, 1.0]])
    agent_from_world = np.linalg.inv(world_from_agent)
    pts = np.concatenate([hist_world[:, :2], np.ones((hist_world.shape[0], 1))], -1)
    local = (agent_from_world @ pts.T).T[:, :2]
    d = hist_world[:, 2] - pose[2]
    return local, np.arctan2(np.sin(d), np.cos(d)), world_from_agent, agent_from_world


def neighbors(pose_all, scene_index, b, radius, max_neighbors):
    """Indices of agent b's neighbours: same scene, closer than radius, nearest first."""
    d = np.linalg.norm(pose_all[:, :2] - pose_all[b, :2], axis=-1)
    cand = [j for j in range(pose_all.shape[0]) if j != b and scene_index[j] == scene_index[b] and d[j] < radius]
    cand.sort(key=lambda j: d[j])
    return cand[:max_neighbors]


def rasterize(pose, boxes, raster_from_agent, H, W):
    """Drivable layer (H, W) of one agent: pixel (px, py) -> agent frame -> world; a pixel is
    blocked when it falls strictly inside one of the scene's axis-aligned boxes (cx, cy, hx, hy)."""
    out = np.ones((H, W), dtype=bool)
    c, s = np.cos(pose[2]), np.sin(pose[2])
    for py in range(H):
        for px in range(W):
            ax = (px - raster_from_agent[0, 2]) / raster_from_agent[0, 0]
            ay = (py - raster_from_agent[1, 2]) / raster_from_agent[1, 1]
            wx = c * ax - s * ay + pose[0]
            wy = s * ax + c * ay + pose[1]
            for (cx, cy, hx, hy) in boxes:
                if abs(wx - cx) < hx and abs(wy - cy) < hy:
                    out[py, px] = False
    return out
