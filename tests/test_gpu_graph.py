"""hipGraph replay of the step loop (TraceSampler.forward(use_graph=True)) must be bit-identical to
the eager launch sequence, including when the next call brings new inputs and new guidance tables
of the same structure (they are copied into the captured plan's tensors)."""
import pytest
import torch

from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler
from trace_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("guided", [False, True])
def test_graph_replay_is_bit_identical_to_eager(guided):
    cfg = ModelConfig()
    s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0")
    aps, N = [3, 2], 2
    for seed in (1, 2, 1):
        batch = syn.make_batch(cfg, aps, seed=seed, extent=(0.9, 0.7, 1.75), dense=True)
        if guided:
            g = syn.merge_guidance(
                syn.target_pos_guidance(batch, aps, weight=30000.0, seed=seed),
                [[{"name": "agent_collision", "weight": 1000.0,
                   "params": {"num_disks": 2, "buffer_dist": 0.2}, "agents": None}] for _ in aps])
            s.set_guidance(g, batch)
        init, steps = syn.make_noise(cfg, sum(aps), N, seed=seed + 10)
        kw = dict(num_samp=N, class_free_guide_w=0.5, guide_clean=True, noise_init=init, noise_steps=steps,
                  n_steps=8, return_guidance_losses=True)
        a = s(batch, **kw)
        b = s(batch, use_graph=True, **kw)
        assert torch.equal(a["pred_traj_scaled"], b["pred_traj_scaled"])
        if guided:
            la, lb = a["guidance_layer_losses"], b["guidance_layer_losses"]
            assert all(torch.equal(x, y) for x, y in zip(la, lb))
            assert torch.equal(a["guidance_plan"].choose_action(la), b["guidance_plan"].choose_action(lb))
    assert len(s._graphs) == 1          # one capture, two replays


def test_graph_replay_with_the_two_stream_row_split():
    """At 4096 rows and more the eager denoiser runs the two halves of its rows on two streams (unet.hip: run_split,
    fork / join by events); a call under stream capture stays on one stream.  Same kernels per row either way: eager
    and replay must agree bit for bit."""
    cfg = ModelConfig()
    s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0")
    aps, N = [16] * 8, 20                       # 2560 trajectories, 5120 denoiser rows with classifier-free guidance
    batch = syn.make_batch(cfg, aps, seed=4)
    s.set_guidance(syn.target_pos_guidance(batch, aps, weight=30000.0, seed=4), batch)
    init, steps = syn.make_noise(cfg, sum(aps), N, seed=14)
    kw = dict(num_samp=N, class_free_guide_w=0.5, guide_clean=True, noise_init=init, noise_steps=steps, n_steps=3)
    a = s(batch, **kw)
    b = s(batch, use_graph=True, **kw)
    c = s(batch, use_graph=True, **kw)
    assert bool(torch.isfinite(a["pred_traj_scaled"]).all())
    assert torch.equal(a["pred_traj_scaled"], b["pred_traj_scaled"])
    assert torch.equal(a["pred_traj_scaled"], c["pred_traj_scaled"])
