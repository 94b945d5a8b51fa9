"""Per-guidance success metrics against vectors from the reference's own guidance_metrics.py
(oracle/gen_golden_metrics.py wrote tests/golden/guidance_metrics.npz by importing it), plus the one class that has
no vector -- the oriented-box collision test, for which the reference asks shapely -- against a brute-force
restatement ("parity unpinned" for that class)."""
import json
import os

import numpy as np
import torch

from trace_amd.guidance_metrics import (AgentCollisionGuidance, boxes_intersect, guidance_metrics_from_config,
                                        GUIDANCE_NAME_TO_METRICS)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "guidance_metrics.npz")


def _load():
    z = np.load(GOLD)
    cfg = json.loads(bytes(z["config_json"]).decode())
    skipped = json.loads(bytes(z["skipped_json"]).decode())
    steps = z["state/centroid"].shape[0]
    image = torch.from_numpy(z["state/image"])
    states = []
    for k in range(steps):
        st = {key: torch.from_numpy(z["state/" + key][k]) for key in
              ("centroid", "yaw", "curr_speed", "extent", "scene_index", "agent_from_world", "raster_from_world")}
        st["image"], st["drivable_map"] = image, image[:, 0]           # layer 0 of the raster (trajdata_utils.py:97-99)
        states.append(st)
    expected = {k[len("expected/"):]: z[k] for k in z.files if k.startswith("expected/")}
    return cfg, skipped, states, expected


def test_metric_keys_follow_the_reference_naming():
    cfg, skipped, _, expected = _load()
    mets = guidance_metrics_from_config(cfg)
    assert set(mets) == set(expected) | set(skipped)
    assert set(GUIDANCE_NAME_TO_METRICS) == {"target_speed", "agent_collision", "map_collision", "target_pos_at_time",
                                             "target_pos", "global_target_pos_at_time", "global_target_pos",
                                             "social_group"}


def test_every_metric_matches_the_reference_on_a_recorded_episode():
    cfg, skipped, states, expected = _load()
    mets = guidance_metrics_from_config(cfg)
    scenes = np.unique(states[0]["scene_index"].numpy())
    for name, want in expected.items():
        m = mets[name]
        for st in states:
            m.add_step(st, scenes)
        got = np.asarray(m.get_episode_metrics(), np.float64)
        assert got.shape == want.shape, name
        assert np.array_equal(np.isnan(got), np.isnan(want)), name      # NaN everywhere but the metric's own scene
        np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=2e-6, atol=1e-6, err_msg=name)
        m.reset()
        m.add_step(states[0], scenes)                                   # usable again after reset
    assert len(expected) == 14


def _brute_force_overlap(ca, cb):
    """Convex polygons overlap unless some edge normal of either separates them (plain loops)."""
    for poly in (ca, cb):
        for i in range(4):
            e = poly[(i + 1) % 4] - poly[i]
            n = np.array([-e[1], e[0]])
            pa, pb = ca @ n, cb @ n
            if pa.max() < pb.min() or pb.max() < pa.min():
                return False
    return True


def test_oriented_box_overlap_against_brute_force():
    rng = np.random.default_rng(3)
    n = 40
    pos = rng.uniform(-8, 8, (n, 2))
    yaw = rng.uniform(-np.pi, np.pi, n)
    ext = np.stack([rng.uniform(1.0, 5.0, n), rng.uniform(0.6, 2.2, n)], -1)
    hit = np.asarray(boxes_intersect(torch.from_numpy(pos), torch.from_numpy(yaw), torch.from_numpy(ext)))
    cor = []
    for i in range(n):
        c, s = np.cos(yaw[i]), np.sin(yaw[i])
        loc = np.array([[1, 1], [1, -1], [-1, -1], [-1, 1]]) * ext[i] / 2
        cor.append(pos[i] + loc @ np.array([[c, s], [-s, c]]))
    want = np.zeros((n, n), bool)
    for i in range(n):
        for j in range(n):
            want[i, j] = i != j and _brute_force_overlap(cor[i], cor[j])
    assert np.array_equal(hit, want)
    assert 0 < want.sum() < n * n // 2                                   # both outcomes occur

    # the episode metric: fraction of the listed agents that were ever in an overlap
    st = {"centroid": torch.from_numpy(pos), "yaw": torch.from_numpy(yaw),
          "extent": torch.from_numpy(np.concatenate([ext, np.ones((n, 1))], -1)),
          "scene_index": torch.zeros(n, dtype=torch.long)}
    agents = [0, 5, 9, 17]
    m = AgentCollisionGuidance(0, {}, agents, 1)
    m.add_step(st, np.array([0]))
    far = dict(st, centroid=torch.from_numpy(pos + 1000.0 * np.arange(n)[:, None]))
    m.add_step(far, np.array([0]))
    assert np.isclose(m.get_episode_metrics()[0], want.any(1)[agents].mean())
