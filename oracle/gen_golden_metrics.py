"""Golden vectors for the per-guidance success metrics -- TEST INFRASTRUCTURE ONLY (runs in the build container).

Imports the reference's own tbsim/utils/guidance_metrics.py through oracle/ref_harness.py, feeds every metric class a
seeded synthetic episode (two scenes, 7 + 5 agents, 12 states; the state dictionaries the closed-loop environment hands
to `add_step`, env_trajdata.py:345-358) and writes the inputs and the reference's `get_episode_metrics()` outputs to
tests/golden/guidance_metrics.npz.  `AgentCollisionGuidance` (oriented boxes) asks shapely, which is absent here: it
has no vector ("parity unpinned" for that one class; tests/test_host_cpu.py checks it against a brute-force
separating-axis restatement instead).

    python oracle/gen_golden_metrics.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402


def synthetic_episode(seed=11, counts=(7, 5), steps=12, H=64, W=64, pix_per_m=2.0):
    rng = np.random.default_rng(seed)
    B = sum(counts)
    scene_index = np.concatenate([np.full(c, 40 + 3 * i) for i, c in enumerate(counts)])   # global ids, consecutive runs
    pos = rng.uniform(-6, 6, (B, 2))
    yaw = rng.uniform(-np.pi, np.pi, B)
    spd = rng.uniform(0.2, 2.0, B)
    extent = np.stack([rng.uniform(0.6, 4.5, B), rng.uniform(0.5, 2.0, B), np.full(B, 1.7)], -1)
    # one raster per agent, centred on its first pose (the plan's frame), layer 0 = drivable
    image = np.zeros((B, 2, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        ang = rng.uniform(0, np.pi)
        band = np.abs((xx - W / 2) * np.sin(ang) - (yy - H / 2) * np.cos(ang)) < rng.uniform(4, 9)
        image[b, 0] = band
        image[b, 1] = rng.uniform(0, 1, (H, W)) < 0.1
    c0, s0 = np.cos(yaw), np.sin(yaw)
    afw = np.zeros((B, 3, 3))
    afw[:, 0, 0], afw[:, 0, 1], afw[:, 1, 0], afw[:, 1, 1], afw[:, 2, 2] = c0, s0, -s0, c0, 1
    afw[:, 0, 2] = -(c0 * pos[:, 0] + s0 * pos[:, 1])
    afw[:, 1, 2] = -(-s0 * pos[:, 0] + c0 * pos[:, 1])
    rfa = np.array([[pix_per_m, 0, W / 4.0], [0, pix_per_m, H / 2.0], [0, 0, 1]])
    rfw = rfa[None] @ afw
    states = []
    for k in range(steps):
        states.append({"centroid": pos.copy(), "yaw": yaw.copy(), "curr_speed": spd.copy(), "extent": extent.copy(),
                       "scene_index": scene_index.copy(),
                       # the environment recomputes agent_from_world per state; the metric must keep the first
                       "agent_from_world": afw.copy() if k == 0 else afw + 0.01 * k,
                       "raster_from_world": rfw.copy(), "image": image})
        yaw = yaw + rng.normal(0, 0.08, B)
        spd = np.clip(spd + rng.normal(0, 0.15, B), 0, None)
        pos = pos + 0.1 * 3 * spd[:, None] * np.stack([np.cos(yaw), np.sin(yaw)], -1)
    return states, counts, pix_per_m


def guidance_config(counts, rng):
    a0, a1 = [0, 2, 3, 6], [1, 4]
    return [
        [{"name": "target_speed", "weight": 1.0, "agents": a0,
          "params": {"target_speed": rng.uniform(0.5, 1.5, len(a0)).tolist()}},
         {"name": "agent_collision", "weight": 1.0, "agents": None, "params": {"num_disks": 1, "buffer_dist": 0.6}},
         {"name": "map_collision", "weight": 1.0, "agents": a0, "params": {"num_points_lw": [10, 10]}},
         {"name": "target_pos", "weight": 1.0, "agents": a0,
          "params": {"target_pos": rng.uniform(-3, 8, (len(a0), 2)).tolist()}},
         {"name": "global_target_pos_at_time", "weight": 1.0, "agents": a0,
          "params": {"target_pos": rng.uniform(-6, 6, (len(a0), 2)).tolist(),
                     "target_time": rng.integers(0, 10, len(a0)).tolist()}}],
        [{"name": "target_pos_at_time", "weight": 1.0, "agents": a1,
          "params": {"target_pos": rng.uniform(-3, 8, (len(a1), 2)).tolist(),
                     "target_time": rng.integers(0, 10, len(a1)).tolist()}},
         {"name": "global_target_pos", "weight": 1.0, "agents": None,
          "params": {"target_pos": rng.uniform(-6, 6, (counts[1], 2)).tolist()}},
         {"name": "social_group", "weight": 1.0, "agents": [0, 2, 3],
          "params": {"leader_idx": 0, "social_dist": 1.5, "cohesion": 0.8}},
         {"name": "agent_collision", "weight": 1.0, "agents": a1, "params": {"num_disks": 1, "buffer_dist": 1.5}},
         {"name": "map_collision", "weight": 1.0, "agents": None, "params": {"num_points_lw": [10, 10]}}],
    ]


def main():
    rh._install_stubs()
    sys.dont_write_bytecode = True
    sys.path.insert(0, rh.REF_ROOT)
    import tbsim.utils.guidance_metrics as gm
    import tbsim.utils.trajdata_utils as tdu

    states, counts, pix_per_m = synthetic_episode()
    tdu.set_global_trajdata_batch_raster_cfg({"include_hist": False, "pixel_size": 1.0 / pix_per_m, "raster_size": 64,
                                              "ego_center": (-0.5, 0.0), "num_sem_layers": 2,
                                              "no_map_fill_value": -1.0, "drivable_layers": [0]})
    cfg = guidance_config(counts, np.random.default_rng(5))
    metrics = gm.guidance_metrics_from_config(cfg)
    skipped = [k for k in metrics if k.startswith("guide_agent_collision_s")]      # shapely polygons
    all_scene_index = np.unique(states[0]["scene_index"])
    out = {}
    for name, met in metrics.items():
        if name in skipped:
            continue
        for st in states:
            met.add_step({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}, all_scene_index)
        out[name] = np.asarray(met.get_episode_metrics(), np.float64)
        print("%-44s %s" % (name, out[name]))
    save = {"expected/" + k: v for k, v in out.items()}
    for key in ("centroid", "yaw", "curr_speed", "extent", "scene_index", "agent_from_world", "raster_from_world"):
        save["state/" + key] = np.stack([st[key] for st in states])
    save["state/image"] = states[0]["image"]
    save["config_json"] = np.frombuffer(json.dumps(cfg).encode(), np.uint8)
    save["skipped_json"] = np.frombuffer(json.dumps(skipped).encode(), np.uint8)
    path = os.path.join(HERE, "..", "tests", "golden", "guidance_metrics.npz")
    np.savez_compressed(path, **save)
    print("wrote", os.path.normpath(path), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
