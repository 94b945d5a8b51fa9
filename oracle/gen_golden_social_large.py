"""Golden vectors for SocialGroupLoss on a group of MORE than 25 members, from the reference itself -- TEST
INFRASTRUCTURE ONLY (build container only; needs /root/reference).

    python -m oracle.gen_golden_social_large      ->  tests/golden/social_group_large.npz

Above 25 members ``torch.cdist`` (tbsim/utils/guidance_loss.py:786) takes its matmul form
|a|^2 + |b|^2 - 2ab, whose rounding noise decides which member is the nearest when two candidates are close;
the HIP kernel restates that form in ATen's operation order (trace_amd/csrc/guidance.hip,
social_neighbor_kernel).  One scene of 30 agents (the group: 28 of them, leader = batch index 2) and one of
2 agents (a small group in the exact form beside it); the reference's ``torch.randint`` / ``torch.rand``
draws are recorded as in gen_golden_social.py.  Inputs are stored in the file: the batch is
``syn.make_batch(cfg, [30, 2], seed=3, extent=(0.9, 0.7, 1.75), dense=True)``.
"""
import os

import numpy as np
import torch

from oracle import ref_harness as rh
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
APS = [30, 2]
N_SAMP = 3


def large_configs():
    sg = lambda w, leader, dist, coh, agents: {"name": "social_group", "weight": w,
                                               "params": {"leader_idx": leader, "social_dist": dist, "cohesion": coh},
                                               "agents": agents}
    members = [i for i in range(30) if i not in (5, 17)]          # 28 of the 30 agents
    return {
        "large_mixed": [[sg(2.0, 2, 1.2, 0.3, members)], [sg(1.0, 30, 0.9, 0.5, None)]],
        "large_nearest": [[sg(1.0, 0, 1.0, 0.0, None)], []],      # all 30, nearest neighbour only
    }


def make_traj(cfg, batch):
    """(B, N, T, 6) seeded trajectories in the agent frames: smooth curves a few metres long, so that members of
    the dense scene come within centimetres of each other at some steps (near-ties of the nearest neighbour)."""
    g = torch.Generator().manual_seed(77)
    B, T = sum(APS), cfg.horizon
    t = torch.linspace(0, 1, T)[None, None, :]
    v = 2.0 + 3.0 * torch.rand(B, N_SAMP, 1, generator=g)
    curve = 0.8 * torch.randn(B, N_SAMP, 1, generator=g)
    x = v * t * 5.2 + 0.05 * torch.randn(B, N_SAMP, T, generator=g)
    y = curve * (t * 5.2) ** 2 * 0.1 + 0.05 * torch.randn(B, N_SAMP, T, generator=g)
    yaw = torch.atan2(2 * curve * t * 5.2 * 0.1, torch.ones_like(t)).expand(B, N_SAMP, T)
    vel = v.expand(B, N_SAMP, T)
    acc = 0.1 * torch.randn(B, N_SAMP, T, generator=g)
    yr = 0.1 * torch.randn(B, N_SAMP, T, generator=g)
    return torch.stack([x, y, vel, yaw, acc, yr], -1).contiguous()


def main():
    ns = rh.load_reference()
    cfg = ModelConfig()
    batch = syn.make_batch(cfg, APS, seed=3, extent=(0.9, 0.7, 1.75), dense=True)
    traj = make_traj(cfg, batch)
    res = {"traj": traj.numpy()}
    real_randint, real_rand = torch.randint, torch.rand
    gen = torch.Generator().manual_seed(2025)
    for name, cfgs in large_configs().items():
        draws = []

        def rec_randint(lo, hi, shape, **kw):
            t = real_randint(lo, hi, shape, generator=gen)
            draws.append(("randint", t.clone()))
            return t

        def rec_rand(*shape, **kw):
            t = real_rand(*shape, generator=gen)
            draws.append(("rand", t.clone()))
            return t

        torch.randint, torch.rand = rec_randint, rec_rand
        try:
            tr = traj.clone().requires_grad_()
            guid = ns.guidance.DiffuserGuidance(cfgs, batch)
            tot, per = guid.compute_guidance_loss(tr, batch)
            gt, = torch.autograd.grad(tot, tr)
        finally:
            torch.randint, torch.rand = real_randint, real_rand
        res["%s/total" % name] = tot.detach().numpy()
        res["%s/grad" % name] = gt.numpy()
        keys = list(per.keys())
        assert len(draws) == 2 * len(keys) and all(d[0] == ("randint", "rand")[i % 2] for i, d in enumerate(draws))
        for gi, k in enumerate(keys):
            res["%s/per/%s" % (name, k)] = per[k].numpy()
            res["%s/randint/%s" % (name, k)] = draws[2 * gi][1].numpy().astype(np.int16)
            res["%s/rand/%s" % (name, k)] = draws[2 * gi + 1][1].numpy()
    np.savez_compressed(os.path.join(OUT, "social_group_large.npz"), **res)
    print("social_group_large.npz:", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
