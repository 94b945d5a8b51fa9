"""Map encoder (row A0): the functional implementation in trace_amd/map_encoder.py against an
independent nn.Module assembly of the same layers (torchvision-style ResNet-18 naming), loaded
from the same state_dict.  PARITY UNPINNED against the reference itself: torchvision is absent, so
the reference's MapEncoder cannot be instantiated here."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from trace_amd.map_encoder import (MapEncoder, has_map_encoder, map_encoder_shapes,
                                   random_map_encoder_state_dict)


class _BasicBlock(nn.Module):
    def __init__(self, cin, c, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, c, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)
        self.downsample = None
        if stride != 1 or cin != c:
            self.downsample = nn.Sequential(nn.Conv2d(cin, c, 1, stride, bias=False), nn.BatchNorm2d(c))

    def forward(self, x):
        o = F.relu(self.bn1(self.conv1(x)))
        o = self.bn2(self.conv2(o))
        return F.relu(o + (x if self.downsample is None else self.downsample(x)))


class _Trunk(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        c0 = 64
        for i, c in enumerate((64, 128, 256, 512), start=1):
            setattr(self, "layer%d" % i, nn.Sequential(_BasicBlock(c0, c, 1 if i == 1 else 2), _BasicBlock(c, c, 1)))
            c0 = c

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        f = []
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
            f.append(x)
        return f


class _DoubleConv(nn.Module):
    def __init__(self, ci, co, mid):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv2d(ci, mid, 3, padding=1), nn.BatchNorm2d(mid), nn.ReLU(),
                                         nn.Conv2d(mid, co, 3, padding=1), nn.BatchNorm2d(co), nn.ReLU())


class _Up(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.conv = _DoubleConv(ci, co, ci // 2)

    def forward(self, x1, x2):
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
        return self.conv.double_conv(torch.cat([x2, x1], 1))


class _ConvBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(64, 64, 1, bias=False), nn.BatchNorm2d(64)
        self.conv2, self.bn2 = nn.Conv2d(64, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64)
        self.conv3, self.bn3 = nn.Conv2d(64, 64, 1, bias=False), nn.BatchNorm2d(64)
        self.shortcut = nn.Sequential(nn.Conv2d(64, 64, 1, bias=False), nn.BatchNorm2d(64))

    def forward(self, x):
        o = F.relu(self.bn1(self.conv1(x)))
        o = F.relu(self.bn2(self.conv2(o)))
        return F.relu(self.bn3(self.conv3(o)) + self.shortcut(x))


class _Decoder(nn.Module):
    def __init__(self, out, hw):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(512, 512, 3, 1, 1, bias=False), nn.ReLU())
        self.up1, self.up2, self.up3 = _Up(768, 256), _Up(384, 128), _Up(192, 64)
        self.conv2 = _ConvBlock()
        self.conv3 = nn.Conv2d(64, out, 1)
        self.out_norm = nn.LayerNorm((out, hw, hw))

    def forward(self, f):
        x = self.conv1(f[3])
        x = self.up3(self.up2(self.up1(x, f[2]), f[1]), f[0])
        return self.out_norm(self.conv3(self.conv2(x)))


class _Heads(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.map_model = _Trunk(cin)


class _Ref(nn.Module):
    def __init__(self, cin, out, hw):
        super().__init__()
        self.encoder_heads = _Heads(cin)
        self.decoder = _Decoder(out, hw)

    def forward(self, x):
        return self.decoder(self.encoder_heads.map_model(x))


def test_functional_map_encoder_matches_module_assembly():
    cin, out, H = 2, 32, 224
    sd = random_map_encoder_state_dict(cin, out, (H, H), seed=3)
    assert has_map_encoder(sd) and set(sd) == set(map_encoder_shapes(cin, out, (H, H)))
    ref = _Ref(cin, out, H // 4).eval()
    missing, unexpected = ref.load_state_dict({k[len("map_encoder."):]: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(1)
    img = (torch.rand(2, cin, H, H, generator=g) > 0.6).float()
    with torch.no_grad():
        want = ref(img)
    glob, got = MapEncoder(sd, "cpu")(img)
    assert glob is None and got.shape == (2, out, H // 4, H // 4)
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 2e-4 * float(want.abs().max())
    # the class-free "unconditional" image (all cond_fill_value) is a valid input too
    _, g2 = MapEncoder(sd, "cpu")(torch.ones_like(img) * 0.5)
    assert torch.isfinite(g2).all() and float((g2[0] - g2[1]).abs().max()) == 0.0


def test_missing_weights_raise():
    sd = random_map_encoder_state_dict(2, 32, (224, 224), seed=0)
    del sd["map_encoder.decoder.conv3.bias"]
    try:
        MapEncoder(sd, "cpu")
    except KeyError as e:
        assert "conv3.bias" in str(e)
    else:
        raise AssertionError("expected KeyError")


# ---- the decoder half against the REFERENCE's own MapGridDecoder (tests/golden/map_decoder.npz) ----------------
def _decoder_fixture():
    """Weights / features rebuilt from their names (oracle/gen_golden_map_decoder.py), checked against the checksums
    the generator recorded next to the reference's output."""
    import numpy as np
    from oracle.gen_golden_map_decoder import decoder_features, decoder_tensor
    from trace_amd.map_encoder import map_encoder_shapes, random_map_encoder_state_dict
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "map_decoder.npz"))
    shapes = map_encoder_shapes(2, 32)
    sd = random_map_encoder_state_dict(2, 32, seed=3)          # trunk weights: unused by decode()
    names = [str(n) for n in gold["names"]]
    # the reference's decoder has exactly the tensors the product expects below `map_encoder.decoder.`
    assert sorted("map_encoder.decoder." + n for n in names) == sorted(k for k in shapes if k.startswith("map_encoder.decoder."))
    digest = 0.0
    for n in names:
        t = decoder_tensor(n, shapes["map_encoder.decoder." + n])
        sd["map_encoder.decoder." + n] = t
        digest += float(t.double().abs().sum())
    feats = decoder_features(2)
    assert abs(digest - float(gold["weight_abs_sum"])) <= 1e-9 * digest
    fsum = sum(float(f.double().abs().sum()) for f in feats)
    assert abs(fsum - float(gold["feat_abs_sum"])) <= 1e-9 * fsum
    return sd, feats, torch.from_numpy(gold["out"])


def test_decoder_half_reproduces_the_reference_map_grid_decoder():
    """`MapEncoder.decode` (torch backend, CPU) against the output of the reference's MapGridDecoder
    (tbsim/models/trace_helpers.py:231-268) on the same weights and features: same ATen ops, 1e-5 of the scale
    (the reference applies BatchNorm as a module, here it is a folded affine)."""
    from trace_amd.map_encoder import MapEncoder
    sd, feats, want = _decoder_fixture()
    got = MapEncoder(sd, device="cpu", backend="torch").decode(feats)
    assert got.shape == want.shape
    err = float((got - want).abs().max())
    assert err <= 1e-5 * float(want.abs().max()), err
