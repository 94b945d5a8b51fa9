"""Map encoder (ResNet-18 + grid decoder on MIOpen) at closed-loop batch sizes: NCHW against the
channels-last path MapEncoder takes from CHANNELS_LAST_FROM images per call.
Usage: python scripts/prof_map_encoder.py [agents]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sd = random_map_encoder_state_dict(2, 32, (224, 224), seed=0)
img = (torch.rand(B, 2, 224, 224, device="cuda:0") > 0.3).float()
outs = {}
for name, from_n in (("nchw", 10 ** 9), ("channels_last", 1)):
    enc = MapEncoder(sd, "cuda:0", channels_last_from=from_n)
    t = []
    for n in (1, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = enc(img)[1]
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / n * 1e3)
    outs[name] = out
    print("%s: first call %.0f ms, steady %.1f ms per call of %d images" % (name, t[0], t[1], B), flush=True)
d = (outs["nchw"] - outs["channels_last"]).abs().max() / outs["nchw"].abs().max()
print("max relative difference between the two layouts: %.2e" % float(d))
