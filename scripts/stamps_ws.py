"""Where the waves of conv_ws_kernel spend their time (diagnostic build -DTRACE_CONV_STAMPS):
    TRACE_CONV_WS=1 TRACE_HIP_LIB=trace_amd/lib/var_stamps.so python scripts/stamps_ws.py 2 8 17
Arguments: conv launch indices (issue order of one forward + input-VJP).  Per workgroup: K wave 0 records the time
it waited at MID / END barriers and wrote output tiles; helper wave 0 the time per activity."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = 10240
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
L.trace_debug_stamps.restype = None; L.trace_debug_stamps.argtypes = [C.c_void_p, C.c_int]
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
pred = torch.empty(rows, 52, 2, device="cuda:0"); gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g); gx = torch.empty(rows, 52, 2, device="cuda:0")
def run():
    _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(), ws.data_ptr(), wsb, 1, None))
    _cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()
L.trace_debug_stamps(None, -1)
for _ in range(3): run()
for target in [int(a) for a in sys.argv[1:]]:
    buf = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda:0")
    L.trace_debug_stamps(buf.data_ptr(), target); run(); L.trace_debug_stamps(None, -1)
    st = buf.cpu().numpy().reshape(-1, 16); st = st[st[:, 0] != 0]
    if st.shape[0] == 0:
        print("launch %d: no stamps (not a conv_ws_kernel launch)" % target); continue
    med = lambda v: float(np.median(v))
    nt, nch = med(st[:, 6]), med(st[:, 7])
    print("launch %d: %d workgroups, %.0f tiles x %.0f chunks each" % (target, st.shape[0], nt, nch))
    print("  K wave : loop %8.0f cycles = %7.0f per tile | MID wait %7.0f  END wait %7.0f  tile write %6.0f (per tile) | prologue %6.0f" % (
        med(st[:, 2]), med(st[:, 2]) / nt, med(st[:, 3]) / nt, med(st[:, 4]) / nt, med(st[:, 5]) / nt, med(st[:, 1])))
    print("  helper : even piece+loads %7.0f  MID wait %7.0f  split+write %7.0f  odd piece %7.0f  END wait %7.0f (per tile)" % tuple(
        med(st[:, 8 + i]) / nt for i in range(5)))
    print("  MFMA cycles per tile and K wave: %d" % (13 * 6 * 16 * 5 * nch))
