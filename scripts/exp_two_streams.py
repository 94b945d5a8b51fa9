"""Experiment: the two halves of the denoiser rows (conditional / unconditional) on two HIP streams, the second one
delayed, so that matrix-bound (L = 13) launches of one half run beside memory-bound (L = 52 / 26) launches of the
other.  Prints ms per forward + input-VJP for: one stream with all rows, two streams in lock-step, two streams with
the second delayed by a fraction of a pass."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = 10240
half = rows // 2
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
pred = torch.empty(rows, 52, 2, device="cuda:0"); gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g); gx = torch.empty(rows, 52, 2, device="cuda:0")
wsb = L.trace_unet_workspace_bytes(s._unet, rows, 1); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda:0")
wsh = L.trace_unet_workspace_bytes(s._unet, half, 1)
wsA = torch.empty(wsh, dtype=torch.uint8, device="cuda:0"); wsB = torch.empty(wsh, dtype=torch.uint8, device="cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both(stream, r0, n, w, wb):
    st = stream.cuda_stream if stream is not None else None
    _cabi.check(L.trace_unet_forward(s._unet, x_in[r0:].data_ptr(), table[r0 // 20:].data_ptr(), 20, 50, n, pred[r0:].data_ptr(), w.data_ptr(), wb, 1, st))
    _cabi.check(L.trace_unet_backward_input(s._unet, gp[r0:].data_ptr(), n, gx[r0:].data_ptr(), w.data_ptr(), wb, st))


def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def one():
    both(None, 0, rows, ws, wsb)


def two(delay_cycles):
    def f():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        if delay_cycles:
            with torch.cuda.stream(s2):
                torch.cuda._sleep(delay_cycles)
        both(s1, 0, half, wsA, wsh)
        both(s2, half, half, wsB, wsh)
        cur.wait_stream(s1); cur.wait_stream(s2)
    return f


print("one stream, %d rows: %.3f ms" % (rows, timed(one)))
ref = gx.clone()
for d in (0, 1_000_000, 2_500_000, 4_000_000, 6_000_000):
    print("two streams, second delayed by %.1f M cycles: %.3f ms" % (d / 1e6, timed(two(d))))
print("same results:", bool(torch.equal(ref, gx)))
