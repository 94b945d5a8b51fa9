"""Golden vectors for SocialGroupLoss from the reference itself -- TEST INFRASTRUCTURE ONLY
(build container only; needs /root/reference).

    python -m oracle.gen_golden_social      ->  tests/golden/social_group.npz

The reference draws ``torch.randint`` / ``torch.rand`` inside the loss
(tbsim/utils/guidance_loss.py:803-804); both are wrapped here so that every draw is recorded next to
the loss / gradient it produced.  Inputs: the trajectory tensor and batch of components.npz
(``loss/traj``; ``syn.make_batch(cfg, [3, 2], seed=1, extent=(0.9, 0.7, 1.75), dense=True)``).
"""
import os

import numpy as np
import torch

from oracle import ref_harness as rh
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def social_configs():
    """name -> guidance config list (two scenes of 3 and 2 agents).  ``leader_idx`` is compared with the
    BATCH index by the reference (guidance_loss.py:775), hence 3 for the first agent of scene 1."""
    sg = lambda w, leader, dist, coh, agents: {"name": "social_group", "weight": w,
                                               "params": {"leader_idx": leader, "social_dist": dist, "cohesion": coh},
                                               "agents": agents}
    return {
        "mixed": [[sg(2.0, 0, 1.5, 0.5, None)], [sg(3.0, 3, 0.8, 0.5, None)]],
        "nearest_only": [[sg(1.0, 1, 1.0, 0.0, None)], []],
        "random_only_sub": [[sg(1.5, 2, 2.0, 1.0, [0, 2])], [sg(0.5, 0, 1.2, 1.0, [0, 1])]],
    }


def main():
    ns = rh.load_reference()
    cfg = ModelConfig()
    aps = [3, 2]
    batch = syn.make_batch(cfg, aps, seed=1, extent=(0.9, 0.7, 1.75), dense=True)
    traj = torch.from_numpy(np.load(os.path.join(OUT, "components.npz"))["loss/traj"])
    res = {}
    real_randint, real_rand = torch.randint, torch.rand
    gen = torch.Generator().manual_seed(2024)
    for name, cfgs in social_configs().items():
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
            res["%s/randint/%s" % (name, k)] = draws[2 * gi][1].numpy()
            res["%s/rand/%s" % (name, k)] = draws[2 * gi + 1][1].numpy()
    np.savez_compressed(os.path.join(OUT, "social_group.npz"), **res)
    print("social_group.npz:", {k: v.shape for k, v in res.items()})


if __name__ == "__main__":
    main()
