"""A/B of the 32 x 32 x 16 conv kernels (trace_conv_m32_mode) against the 16 x 16 x 32 ones on one denoiser pass
(forward + input-VJP, bench shape): agreement of the outputs and ms per pass of each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g)
res = {}
for mode in (0, 1, 0, 1):
    L.trace_conv_m32_mode(mode)
    pred = torch.empty(rows, 52, 2, device="cuda:0"); gx = torch.empty(rows, 52, 2, device="cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(iters):
        if i == 1:
            e0.record()
        _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(), ws.data_ptr(), wsb, 1, None))
        _cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
    e1.record(); torch.cuda.synchronize()
    print("m32=%d: %.3f ms per pass" % (mode, e0.elapsed_time(e1) / (iters - 1)), flush=True)
    res[mode] = (pred.clone(), gx.clone())
for k, name in ((0, "pred"), (1, "grad_x")):
    a, b = res[0][k], res[1][k]
    print("%s: max |diff| / max |ref| = %.3e, finite %s" % (name, float((a - b).abs().max() / a.abs().max()), bool(torch.isfinite(b).all())))
