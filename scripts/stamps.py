"""Phase timing of conv_gemm_kernel workgroups from in-kernel s_memtime stamps (diagnostic build).

    # build the stamped library (not part of the product build):
    #   hipcc ... -DTRACE_CONV_STAMPS ... -o trace_amd/lib/libtrace_hip_stamps.so   
    TRACE_HIP_LIB=trace_amd/lib/libtrace_hip_stamps.so python scripts/stamps.py 17 45

Arguments: conv launch indices (in issue order of one forward + input-VJP) to stamp.
Per workgroup the kernel records: 0 start, 1 first K loop starts, 2.. end of each chunk's K loop,
10 accumulators in LDS, 11 end, 12 HW_ID, 13 XCC_ID.
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
gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g)
gx = torch.empty(rows, 52, 2, device="cuda:0")


def run():
    _cabi.check(L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 50, rows, pred.data_ptr(),
                                     ws.data_ptr(), wsb, 1, None))
    _cabi.check(L.trace_unet_backward_input(s._unet, gp.data_ptr(), rows, gx.data_ptr(), ws.data_ptr(), wsb, None))
    torch.cuda.synchronize()


L.trace_debug_stamps(None, -1)
for _ in range(3):
    run()                                   # warm-up: clocks, caches
for target in [int(a) for a in sys.argv[1:]]:
    buf = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda:0")
    L.trace_debug_stamps(buf.data_ptr(), target)
    run()
    L.trace_debug_stamps(None, -1)
    st = buf.cpu().numpy().reshape(-1, 16)
    st = st[st[:, 0] != 0]
    n = st.shape[0]
    hw, xcc = st[:, 12], st[:, 13] & 15
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15))
    nchunk = int((st[:, 2:6] != 0).sum(1).max())
    t0 = st[:, 0]
    total = st[:, 11] - t0
    prol = st[:, 1] - t0
    kend = st[:, 1 + nchunk]
    kphase = kend - st[:, 1]
    e0 = st[:, 10] - kend
    epi = st[:, 11] - st[:, 10]
    print("launch %d: %d workgroups, %d chunks" % (target, n, nchunk))
    for name, v in (("total", total), ("prologue", prol), ("K phases (all chunks)", kphase),
                    ("acc->LDS", e0), ("epilogue rest", epi)):
        print("  %-24s median %8.0f   p10 %8.0f   p90 %8.0f cycles" % (name, np.median(v), np.percentile(v, 10),
                                                                   np.percentile(v, 90)))
    print("    prologue: start -> zero fill done %8.0f | -> first chunk in LDS (load wait + writes) %8.0f | -> K starts %8.0f" % (
        np.median(st[:, 14] - t0), np.median(st[:, 15] - st[:, 14]), np.median(st[:, 1] - st[:, 15])))
    if (st[:, 9] != 0).all():
        print("    epilogue: statistics/tables %8.0f | normalise+store pass %8.0f" % (
            np.median(st[:, 9] - st[:, 10]), np.median(st[:, 11] - st[:, 9])))
    if (st[:, 6] != 0).all() and (st[:, 8] != 0).all():
        print("      statistics: row moments %8.0f | group sums %8.0f | tables+stats %8.0f | barrier %8.0f" % (
            np.median(st[:, 6] - st[:, 10]), np.median(st[:, 7] - st[:, 6]), np.median(st[:, 8] - st[:, 7]),
            np.median(st[:, 9] - st[:, 8])))
    for c in range(nchunk):
        a = st[:, 2 + c] - (st[:, 1 + c])
        print("    chunk %d (K loop + next switch) median %8.0f" % (c, np.median(a)))
    # co-residency: for every workgroup, how much of its lifetime overlapped the K phases / the
    # non-K phases of the other workgroup on the same CU (same XCD => same s_memtime clock)
    order = np.argsort(t0)
    by_cu = {}
    for i in order:
        by_cu.setdefault(int(cu[i]), []).append(i)
    both_k = one_k = none_k = 0.0
    gaps = []
    for c, idx in by_cu.items():
        ev = []
        for i in idx:
            ev.append((st[i, 1], +1))        # K phase starts
            ev.append((kend[i], -1))         # K phase ends
        ev.sort()
        lo, hi = min(t0[idx]), max(st[idx, 11])
        cur, last = 0, lo
        for t, d in ev:
            span = t - last
            if cur >= 2:
                both_k += span
            elif cur == 1:
                one_k += span
            else:
                none_k += span
            cur += d
            last = t
        none_k += hi - last
        # slot refill latency: time between a workgroup's end and the next start on this CU
        ends = sorted(st[idx, 11])
        starts = sorted(t0[idx])[2:]
        for e, sst in zip(ends, starts):
            gaps.append(sst - e)
    tot = both_k + one_k + none_k
    print("  per-CU time: both workgroups in K %.1f%%, exactly one %.1f%%, none %.1f%%" % (
        100 * both_k / tot, 100 * one_k / tot, 100 * none_k / tot))
    if gaps:
        print("  slot refill gap (end of a workgroup -> start of the next on the CU): median %.0f cycles" %
              np.median(gaps))
    mf = 13 * 4 * 32
    print("  MFMA cycles per k-block per wave = %d" % mf)
