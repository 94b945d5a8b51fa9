"""Diagnostic: diff the denoiser workspace (all activations) between two identical calls."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trace_amd import synthetic as syn
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler
from trace_amd.weights import random_state_dict

DEV = "cuda:0"
cfg = ModelConfig()
sd = random_state_dict(cfg, 0)
aps = [32] * 8
N = 20
batch = syn.make_batch(cfg, aps, seed=100, max_neighbors=16)
B = sum(aps)
g = torch.Generator().manual_seed(5)
x = (torch.randn(B * N, cfg.horizon, 6, generator=g) * 0.7).to(DEV)
s = TraceSampler(cfg, sd, DEV)
bd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
aux = s.get_aux_info(bd, include_class_free_cond=True)
R = B * N
# buffer plan for save=0 (unet.hip trace_unet_create)
H, d0, d1, d2 = 52, 64, 128, 256
plan = [("cat1", 2 * d1, 26), ("cat2", 2 * d2, 13), ("ds0", d0, 26), ("ds1", d1, 13), ("up1", d0, 52), ("f", d0, 52)]
blocks = [(38, 64, 52), (64, 64, 52), (64, 128, 26), (128, 128, 26), (128, 256, 13), (256, 256, 13), (256, 256, 13),
          (256, 256, 13), (512, 128, 13), (128, 128, 13), (256, 64, 26), (64, 64, 26)]
for i, (ci, co, L) in enumerate(blocks):
    plan.append(("b%d.h1" % i, co, L))
    plan.append(("b%d.out" % i, co, L))
    if ci != co:
        plan.append(("b%d.res" % i, co, L))
offs, o = [], 0
for name, C, L in plan:
    offs.append(o)
    n = C * L * R
    o += (n + 63) // 64 * 64
ws = []
for rep in range(2):
    out = s.denoise(x, 60, aux, N, w=0.0, save=False)
    torch.cuda.synchronize()
    ws.append(out[3][0].view(torch.float32).clone())
print("ws floats", ws[0].numel(), "plan floats", o)
neq = (ws[0] != ws[1]) & ~(torch.isnan(ws[0]) & torch.isnan(ws[1]))
idx = neq.nonzero().flatten().cpu()
print("differing floats:", idx.numel())
first = None
for (name, C, L), off in zip(plan, offs):
    m = (idx >= off) & (idx < off + C * L * R)
    k = int(m.sum())
    if k:
        loc = idx[m] - off
        row, c, l = loc // (C * L), (loc // L) % C, loc % L
        print("%-8s C=%d L=%d: %d differing; rows %d; row%%8 %s; c//16 %s; l %s" % (
            name, C, L, k, row.unique().numel(), sorted(collections.Counter((row % 8).tolist()).items()),
            sorted(collections.Counter((c // 16).tolist()).items()), sorted(collections.Counter(l.tolist()).items())[:30]))
        if first is None:
            first = name
            r0 = int(row[0])
            mm = row == r0
            print("   row", r0, ": channels", sorted(set(c[mm].tolist()))[:70], "positions", sorted(set(l[mm].tolist())))
            a = ws[0][off + r0 * C * L: off + (r0 + 1) * C * L].view(C, L).cpu()
            b = ws[1][off + r0 * C * L: off + (r0 + 1) * C * L].view(C, L).cpu()
            cc = int(c[mm][0])
            print("   run0 ch", cc, a[cc].tolist()[:13])
            print("   run1 ch", cc, b[cc].tolist()[:13])
