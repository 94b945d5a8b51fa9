"""Diagnostic: run the same guided step sequence / denoiser call twice and report bit differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler
from trace_amd.weights import random_state_dict

DEV = "cuda:0"
cfg = ModelConfig()
sd = random_state_dict(cfg, 0)
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
aps = [32] * scenes
N = 20
batch = syn.make_batch(cfg, aps, seed=100, max_neighbors=16)
guidance = syn.target_pos_guidance(batch, aps, weight=30000.0, seed=100)
B = sum(aps)
g = torch.Generator().manual_seed(5)
x = (torch.randn(B * N, cfg.horizon, 6, generator=g) * 0.7).to(DEV)
s = TraceSampler(cfg, sd, DEV)
s.set_guidance(guidance, batch)
bd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
aux = s.get_aux_info(bd, include_class_free_cond=True)
for w in (0.0, 0.5):
    outs = []
    for rep in range(3):
        x0, pred = s.denoise(x, 60, aux, N, w=w, save=False)[:2]
        outs.append(pred.clone())
    for rep in (1, 2):
        d = (outs[rep] - outs[0]).abs()
        print("denoise w=%.1f rep %d: max diff %.3e, n differing %d of %d, rows differing %d" % (
            w, rep, d.max().item(), int((d > 0).sum()), d.numel(), int((d.flatten(1).max(1).values > 0).sum())))
        if d.max() > 0:
            rows = (d.flatten(1).max(1).values > 0).nonzero().flatten()
            print("   first differing rows:", rows[:20].tolist(), " last:", rows[-5:].tolist())
        if d.max() > 0 and rep == 1 and w == 0.5:
            import collections
            rws = rows.tolist()
            print("   row%8 hist:", sorted(collections.Counter(r % 8 for r in rws).items()))
            print("   (row//8)%8 hist:", sorted(collections.Counter((r // 8) % 8 for r in rws).items()))
            print("   row//2048 hist:", sorted(collections.Counter(r // 2048 for r in rws).items()))
            r0 = rws[0]
            dd = d[r0]
            print("   row", r0, "differing positions:", (dd.max(1).values > 0).nonzero().flatten().tolist())
