"""Map encoder (ResNet-18 trunk + MapGridDecoder, 224 x 224 rasters) at closed-loop batch sizes: the library's own
kernels (backend "hip", csrc/conv2d.hip) against the same network on PyTorch-ROCm ops (backend "torch": MIOpen NCHW,
and with `cl` also MIOpen channels-last, which costs minutes of one-time kernel compilation).
Usage: python scripts/prof_map_encoder.py [images] [channels] [cl]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2
with_cl = len(sys.argv) > 3 and sys.argv[3] == "cl"
sd = random_map_encoder_state_dict(C, 32, (224, 224), seed=0)
img = (torch.rand(B, C, 224, 224, device="cuda:0") > 0.3).float()


def timed(enc):
    t = []
    for n in (1, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = enc(img)[1]
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / n * 1e3)
    return out, t


# algorithmic FLOPs of one image: every convolution of the network, 2 * cin * cout * k * k * Ho * Wo
enc = MapEncoder(sd, "cuda:0", backend="hip")
flops = [2.0 * C * 49 * 64 * 112 * 112]
conv0 = enc._conv


def counting(n, L, src, dst, **kw):
    y = dst[0]
    flops.append(2.0 * L["cin"] * L["cout"] * L["k"] * L["k"] * y.shape[1] * y.shape[2])
    return conv0(n, L, src, dst, **kw)


enc._conv = counting
enc.HIP_CHUNK = 1
enc(img[:1])
enc._conv = conv0
enc.HIP_CHUNK = MapEncoder.HIP_CHUNK
enc._hip_bufs = {}
gf = sum(flops) / 1e9
print("algorithmic work: %.2f GFLOP per image (%d convolutions)" % (gf, len(flops)), flush=True)
outs = {}
outs["hip"], t = timed(enc)
print("hip: first call %.0f ms, steady %.1f ms per call of %d images = %.0f images/s, %.1f TFLOP/s fp32 results" % (
    t[0], t[1], B, B / t[1] * 1e3, gf * B / t[1]), flush=True)
del enc
torch.cuda.empty_cache()
for name, from_n in (("torch_nchw", 10 ** 9),) + ((("torch_channels_last", 1),) if with_cl else ()):
    e2 = MapEncoder(sd, "cuda:0", channels_last_from=from_n, backend="torch")
    outs[name], t = timed(e2)
    print("%s: first call %.0f ms, steady %.1f ms per call of %d images = %.1f TFLOP/s" % (name, t[0], t[1], B, gf * B / t[1]),
          flush=True)
    d = (outs["hip"] - outs[name]).abs().max() / outs[name].abs().max()
    print("  max |hip - %s| / max|%s| = %.2e" % (name, name, float(d)), flush=True)
    del e2
    torch.cuda.empty_cache()
