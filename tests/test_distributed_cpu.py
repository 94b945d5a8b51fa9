"""The N>1 path on CPU: two gloo ranks shard the scenes of one batch (trace_amd/distributed.py),
each runs the sampler stand-in on its own scenes, and the one exchange step (all_gather of selected
plans / scores) must reassemble exactly what a single process computes.  The per-rank compute here
is the CPU oracle (no GPU in this test); on GPUs the same host code drives the HIP sampler."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

APS = [3, 2, 1]
N = 3
N_STEPS = 3


def _inputs():
    from trace_amd import synthetic as syn
    from trace_amd.config import ModelConfig
    from trace_amd.weights import random_state_dict
    cfg = ModelConfig()
    sd = random_state_dict(cfg, 0)
    batch = syn.make_batch(cfg, APS, seed=2, extent=(0.9, 0.7, 1.75), dense=True)
    guidance = syn.merge_guidance(
        syn.target_pos_guidance(batch, APS, weight=300.0, seed=5),
        [[{"name": "agent_collision", "weight": 10.0, "params": {"num_disks": 2, "buffer_dist": 0.2},
           "agents": None}] for _ in APS])
    init, steps = syn.make_noise(cfg, sum(APS), N, seed=3)
    return cfg, sd, batch, guidance, init, steps


def _run(cfg, sd, batch, guidance, init, steps):
    from oracle import trace_oracle as orc
    o = orc.OracleSampler(cfg, sd)
    o.set_guidance(guidance)
    out = o.sample(batch, N, 0.0, True, True, noise_init=init, noise_steps=steps, n_steps=N_STEPS,
                   return_diffusion=False)
    B = out["positions"].shape[0]
    idx = orc.choose_action_from_guidance(B, o.guidance.names(), out["guide_losses"])
    scores = torch.nan_to_num(torch.stack(list(out["guide_losses"].values()), -1), nan=0.0).sum(-1)
    ar = torch.arange(B)
    return out["positions"][ar, idx].contiguous(), out["yaws"][ar, idx].contiguous(), scores, idx


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from trace_amd.distributed import gather_actions, shard_scenes, slice_batch, slice_guidance
        cfg, sd, batch, guidance, init, steps = _inputs()
        rng = shard_scenes(APS, world)[rank]
        sub, (a0, a1) = slice_batch(batch, APS, rng)
        # the noise tape is indexed by GLOBAL agent, so results do not depend on the rank count
        init_r = init[a0:a1]
        steps_r = steps.reshape(steps.shape[0], sum(APS), N, *steps.shape[2:])[:, a0:a1].reshape(
            steps.shape[0], (a1 - a0) * N, *steps.shape[2:])
        local = _run(cfg, sd, sub, slice_guidance(guidance, rng), init_r, steps_r)
        pos, yaw, sc, idx = gather_actions(*local)
        if rank == 0:
            q.put((pos.numpy(), yaw.numpy(), sc.numpy(), idx.numpy(), [a0, a1]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_rank_scene_sharding_reassembles_the_single_process_result():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
    try:
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    pos, yaw, sc, idx, rng0 = q.get()
    assert rng0 == [0, 3]                                   # rank 0 owns scene 0 (3 agents), rank 1 scenes 1-2
    ref = _run(*_inputs())
    assert pos.shape == tuple(ref[0].shape) and idx.shape == tuple(ref[3].shape)
    # per-scene arithmetic is identical; only the batch size of the (row-independent) denoiser differs
    assert (torch.from_numpy(pos) - ref[0]).abs().max() < 2e-5
    assert (torch.from_numpy(yaw) - ref[1]).abs().max() < 2e-5
    assert torch.allclose(torch.from_numpy(sc), ref[2], rtol=1e-4, atol=1e-6)
    assert torch.equal(torch.from_numpy(idx), ref[3])


def test_all_gather_ragged_is_identity_without_a_process_group():
    from trace_amd.distributed import all_gather_ragged
    t = torch.arange(6.0).reshape(3, 2)
    assert all_gather_ragged(t) is t
