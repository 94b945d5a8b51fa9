"""Runs a few denoiser forward (+ input-VJP) passes at bench shape, for rocprofv3 counter passes."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bwd = (sys.argv[3] != "fwd") if len(sys.argv) > 3 else True
# "nosplit": keep every launch whole (calls of 4096+ rows otherwise run their two row halves on two streams, and a
# counter pass would then report per-HALF launches): an open per-launch timing profile switches the split off
nosplit = len(sys.argv) > 4 and sys.argv[4] == "nosplit"
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
pred = torch.empty(rows, 52, 2, device="cuda:0"); gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g); gx = torch.empty(rows, 52, 2, device="cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if nosplit:
    _cabi.check(L.trace_profile_begin(200 * iters))
for i in range(iters):
    if i == 1:
        e0.record()          # the first pass warms up (kernel attributes, caches)
    _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(), ws.data_ptr(), wsb, 1, None))
    if bwd:
        _cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
e1.record()
torch.cuda.synchronize()
if nosplit:
    import ctypes
    ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _cabi.check(L.trace_profile_end(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
if iters > 1:
    print("ms per pass (fwd%s): %.3f" % (" + input-VJP" if bwd else "", e0.elapsed_time(e1) / (iters - 1)))
print("done", float(pred.abs().mean()), float(gx.abs().mean()) if bwd else 0)
