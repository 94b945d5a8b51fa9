"""Dumps the denoiser's prediction and input gradient on seeded inputs (A/B of two library builds bit for bit):
    TRACE_HIP_LIB=... python scripts/dump_unet.py ROWS OUT.npz"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = int(sys.argv[1]); out = sys.argv[2]
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
nc = (rows + 19) // 20
cond = torch.randn(nc, 256, device="cuda:0", generator=g); table = torch.empty(nc, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), nc, table.data_ptr(), None))
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1); ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda:0")
pred = torch.empty(rows, 52, 2, device="cuda:0"); gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g); gx = torch.empty(rows, 52, 2, device="cuda:0")
_cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(), ws.data_ptr(), wsb, 1, None))
_cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
torch.cuda.synchronize()
np.savez(out, pred=pred.cpu().numpy(), gx=gx.cpu().numpy(), ws=ws.cpu().numpy())
print("dumped", rows, float(pred.abs().mean()), float(gx.abs().mean()))
