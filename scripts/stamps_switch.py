"""Chunk-switch timing of conv_bf3_kernel from in-kernel stamps (diagnostic build with -DTRACE_CONV_STAMPS
-DTRACE_STAMP_SWITCH): for chunk 1 of the stamped launch, wave 0's view of
  K loop of chunk 0 ends (2) -> every wave arrived at the barrier (4) -> split + LDS writes + barrier (5) -> K loop of chunk 1 ends (3).

    TRACE_HIP_LIB=trace_amd/lib/var_sw.so python scripts/stamps_switch.py 17
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi                                   # noqa: E402
from trace_amd.config import ModelConfig                      # noqa: E402
from trace_amd.sampler import TraceSampler, C_PAD             # noqa: E402
from trace_amd.weights import random_state_dict               # noqa: E402

rows = 10240
cfg = ModelConfig()
s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0")
L = s.lib
L.trace_debug_stamps.restype = None
L.trace_debug_stamps.argtypes = [C.c_void_p, C.c_int]
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0")
x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
pred = torch.empty(rows, 52, 2, device="cuda:0")
gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g)
gx = torch.empty(rows, 52, 2, device="cuda:0")


def run():
    _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(),
                                     ws.data_ptr(), wsb, 1, None))
    _cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()


L.trace_debug_stamps(None, -1)
for _ in range(3):
    run()
for target in [int(a) for a in sys.argv[1:]]:
    buf = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda:0")
    L.trace_debug_stamps(buf.data_ptr(), target)
    run()
    L.trace_debug_stamps(None, -1)
    st = buf.cpu().numpy().reshape(-1, 16)
    st = st[(st[:, 0] != 0) & (st[:, 5] != 0)]
    print("launch %d: %d workgroups" % (target, st.shape[0]))
    for name, v in (("K loop of chunk 0 (1 -> 2)", st[:, 2] - st[:, 1]),
                    ("wait for the other waves (2 -> 4)", st[:, 4] - st[:, 2]),
                    ("split + LDS writes + barrier (4 -> 5)", st[:, 5] - st[:, 4]),
                    ("K loop of chunk 1 (5 -> 3)", st[:, 3] - st[:, 5]),
                    ("whole life (0 -> 11)", st[:, 11] - st[:, 0])):
        print("  %-40s median %7.0f   p10 %7.0f   p90 %7.0f cycles" % (name, np.median(v), np.percentile(v, 10),
                                                                  np.percentile(v, 90)))
