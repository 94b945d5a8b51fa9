"""Experiment: independent row chains on separate HIP streams WITHOUT a join per pass (the product joins its two row
halves every DDPM step).  Scene-disjoint halves of a batch are independent for the whole 100-step loop, so chains with a
constant phase offset could keep matrix-bound (L = 13) launches of one chain beside memory-bound (L = 52 / 26) launches
of another.  Prints ms per 10240-row-equivalent pass (forward + input-VJP)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict
rows = 10240
cfg = ModelConfig(); s = TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0"); L = s.lib
g = torch.Generator(device="cuda:0").manual_seed(0)
x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0"); x_in[:, :38] = torch.randn(rows, 38, 52, device="cuda:0", generator=g)
cond = torch.randn(rows // 20, 256, device="cuda:0", generator=g)
table = torch.empty(rows // 20, s.cond_width, device="cuda:0")
_cabi.check(L.trace_unet_cond_table(s._unet, cond.data_ptr(), rows // 20, table.data_ptr(), None))
pred = torch.empty(rows, 52, 2, device="cuda:0"); gp = torch.randn(rows, 52, 2, device="cuda:0", generator=g); gx = torch.empty(rows, 52, 2, device="cuda:0")
N = 12


def both(st, r0, n, w, wb):
    _cabi.check(L.trace_unet_forward(s._unet, x_in[r0:].data_ptr(), table[r0 // 20:].data_ptr(), 20, 50, n, pred[r0:].data_ptr(), w.data_ptr(), wb, 1, st))
    _cabi.check(L.trace_unet_backward_input(s._unet, gp[r0:].data_ptr(), n, gx[r0:].data_ptr(), w.data_ptr(), wb, st))


def run(nchains, offset_cycles):
    n = rows // nchains
    wb = L.trace_unet_workspace_bytes(s._unet, n, 1)
    wss = [torch.empty(wb, dtype=torch.uint8, device="cuda:0") for _ in range(nchains)]
    streams = [torch.cuda.Stream() for _ in range(nchains)] if nchains > 1 else [torch.cuda.current_stream()]
    def go(iters):
        cur = torch.cuda.current_stream()
        for c, stm in enumerate(streams):
            if stm is not cur:
                stm.wait_stream(cur)
                if offset_cycles and c:
                    with torch.cuda.stream(stm):
                        torch.cuda._sleep(offset_cycles * c)
        for _ in range(iters):
            for c, stm in enumerate(streams):
                both(stm.cuda_stream if stm is not cur else None, c * n, n, wss[c], wb)
        for stm in streams:
            if stm is not cur:
                cur.wait_stream(stm)
    go(2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(N); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


print("1 chain (product: two row halves joined per call): %.3f ms" % run(1, 0))
ref = gx.clone()
for nch in (2, 4):
    for off in (0, 500_000, 1_500_000, 3_000_000):
        print("%d free chains, offset %.1f M cycles per chain: %.3f ms" % (nch, off / 1e6, run(nch, off)))
print("same results:", bool(torch.equal(ref, gx)))
