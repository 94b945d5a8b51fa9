"""Scene-wise sharding of a batch across ranks and the one exchange step of the sampler.

Scenes are the unit of work: agent-agent collision couples the agents
of a scene and sample selection can be scene-level
(/root/reference/tbsim/utils/guidance_loss.py:31-37,252-262), so a scene never straddles two
ranks and nothing is exchanged inside the 100-step loop.  After the loop every rank holds final
actions for its own scenes; one all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests)
assembles the per-sample guidance scores and the selected trajectories on every rank.
"""
import torch
import torch.distributed as dist


def shard_scenes(agents_per_scene, world_size):
    """Contiguous scene ranges per rank, balanced by agent count.
    Returns a list (len world_size) of (scene_begin, scene_end)."""
    n = len(agents_per_scene)
    total = float(sum(agents_per_scene))
    bounds = [0]
    acc = 0.0
    s = 0
    for r in range(1, world_size):
        target = total * r / world_size
        while s < n and acc + agents_per_scene[s] / 2.0 <= target:
            acc += agents_per_scene[s]
            s += 1
        # leave at least one scene for each remaining rank when possible
        s = min(s, max(n - (world_size - r), bounds[-1]))
        bounds.append(s)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def slice_batch(batch, agents_per_scene, scene_range):
    """Sub-batch holding the agents of scenes [begin, end)."""
    b, e = scene_range
    a0 = int(sum(agents_per_scene[:b]))
    a1 = int(sum(agents_per_scene[:e]))
    n_agents = int(sum(agents_per_scene))
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n_agents:
            out[k] = v[a0:a1]
        else:
            out[k] = v
    return out, (a0, a1)


def slice_guidance(guidance_config_list, scene_range):
    b, e = scene_range
    return None if guidance_config_list is None else guidance_config_list[b:e]


def all_gather_ragged(t, group=None):
    """All-gather of tensors that differ in dim 0 (agents per rank): pad to the max, gather,
    trim.  One collective for the sizes, one for the payload."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    ws = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.long)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def gather_actions(positions, yaws, scores, act_idx, group=None):
    """The exchange step: selected plan (B_local,T,2)/(B_local,T,1), per-sample summed guidance
    scores (B_local,N) and chosen sample index (B_local) from every rank, concatenated in rank
    (= scene) order."""
    return (all_gather_ragged(positions, group), all_gather_ragged(yaws, group),
            all_gather_ragged(scores, group), all_gather_ragged(act_idx, group))
