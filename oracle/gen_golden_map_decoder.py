"""Golden vectors for the decoder half of the map encoder -- TEST INFRASTRUCTURE ONLY (build container).

Runs the REFERENCE's own `MapGridDecoder` (tbsim/models/trace_helpers.py:231-268, with `Up` / `DoubleConv` /
`ConvBlock` of tbsim/models/base_models.py:169-260 under it; pure torch, importable through oracle/ref_harness.py)
in evaluation mode on seeded ResNet-18-shaped features, with seeded weights and non-trivial BatchNorm statistics, and
writes tests/golden/map_decoder.npz: the reference's output, the reference's state_dict key names and two checksums.
Weights and features are functions of their NAME (`decoder_tensor`, `decoder_features`: a generator seeded with the
name's crc32), so the tests rebuild them and check the checksums instead of carrying 30 MB of weights.  The ResNet-18 trunk itself needs torchvision (absent) and stays
unpinned.

    python oracle/gen_golden_map_decoder.py
"""
import os
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_harness import load_reference          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "map_decoder.npz")


def _gen(name):
    return torch.Generator().manual_seed(zlib.crc32(name.encode()))


def decoder_tensor(name, shape):
    """Seeded value of the decoder parameter / buffer `name` (key below `map_encoder.decoder.`): a function of the
    name alone, so that the test rebuilds the weights instead of storing 30 MB of them."""
    g = _gen(name)
    if name.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if name.endswith("running_mean"):
        return 0.1 * torch.randn(shape, generator=g)
    if "out_norm" in name or (len(shape) == 1 and name.endswith("weight")):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 1:
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = shape[1] * shape[2] * shape[3]
    return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5


def decoder_features(batch):
    """Seeded post-ReLU features of the shapes ResNet-18's layer1..layer4 emit for a 224 x 224 raster."""
    return [torch.relu(torch.randn(batch, c, hw, hw, generator=_gen("feat%d" % i)))
            for i, (c, hw) in enumerate(((64, 56), (128, 28), (256, 14), (512, 7)), start=1)]


def main():
    ns = load_reference()
    B, gd = 2, 32
    # what MapEncoder.__init__ passes (trace_helpers.py:205-211): layer4 shape, layer1..3 channels
    dec = ns.MapGridDecoder(input_shape=(512, 7, 7), encoder_channels=[64, 128, 256], output_channel=gd,
                            batchnorm=True)
    digest = 0.0
    with torch.no_grad():
        for name, t in dec.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            t.copy_(decoder_tensor(name, tuple(t.shape)))
            digest += float(t.double().abs().sum())
    dec.eval()
    feats = decoder_features(B)
    with torch.no_grad():
        out = dec(feat_to_decode=feats[3], encoder_feats=feats[:3])
    assert out.shape == (B, gd, 56, 56)
    names = [n for n in dec.state_dict() if not n.endswith("num_batches_tracked")]
    np.savez_compressed(OUT, out=out.numpy(), weight_abs_sum=np.float64(digest),
                        feat_abs_sum=np.float64(sum(float(f.double().abs().sum()) for f in feats)),
                        names=np.array(names))
    print("wrote", OUT, out.shape, "%d tensors, %.2f MB" % (len(names), os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main()
