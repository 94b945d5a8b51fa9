"""Per-slice / per-tile timing inside the K loop of conv_bf3_kernel (diagnostic build -DTRACE_CONV_STAMPS
-DTRACE_STAMP_SLICES): wave 0 of every workgroup stamps s_memtime at the start of each (tap, k-slice) of chunks 1
and 2 and at each row tile of the third slice of chunk 1.

    STAMP_ROWS=320 TRACE_HIP_LIB=trace_amd/lib/var_slices.so python scripts/stamps_slices.py 14 15
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

rows = int(os.environ.get("STAMP_ROWS", "10240"))
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


def run():
    _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(),
                                     ws.data_ptr(), wsb, 1, None))
    torch.cuda.synchronize()


L.trace_debug_stamps(None, -1)
for _ in range(3):
    run()
for target in [int(a) for a in sys.argv[1:]]:
    buf = torch.zeros(131072 + 8192 * 64, dtype=torch.int64, device="cuda:0")
    L.trace_debug_stamps(buf.data_ptr(), target)
    run()
    L.trace_debug_stamps(None, -1)
    a = buf.cpu().numpy()
    st = a[:131072].reshape(-1, 16)
    sl = a[131072:].reshape(-1, 64)
    live = st[:, 0] != 0
    sl = sl[live]
    print("launch %d: %d workgroups" % (target, sl.shape[0]))
    for name, base in (("chunk 1", 0), ("chunk 2", 16)):
        t = sl[:, base:base + 11]
        if (t[:, :11] == 0).any():
            print("  %s: not stamped" % name)
            continue
        d = np.diff(t, axis=1)
        print("  %s: cycles per slice (median over workgroups): %s | total %d" % (
            name, " ".join("%5d" % v for v in np.median(d, axis=0)), np.median(t[:, 10] - t[:, 0])))
    t = sl[:, 32:45]
    if not (t == 0).any():
        d = np.diff(t, axis=1)
        print("  chunk 1, slice 2: cycles per row tile: %s" % " ".join("%4d" % v for v in np.median(d, axis=0)))
