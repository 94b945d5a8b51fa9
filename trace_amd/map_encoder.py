"""Map encoder of TRACE's conditioning (row A0 of SURVEY.md section 8): rasterised map image
(B, C, 224, 224) -> `map_grid_feat` (B, 32, 56, 56).

Structure restated from the reference
  * MapEncoder / MapGridDecoder      /root/reference/tbsim/models/trace_helpers.py:176-268
  * RasterizedMapEncoder (ResNet-18 trunk with a C-channel 7x7 stem)
                                      /root/reference/tbsim/models/base_models.py:402-491
  * Up / DoubleConv / ConvBlock       /root/reference/tbsim/models/base_models.py:169-260
and from torchvision's ResNet-18 (BasicBlock x [2,2,2,2]), whose layout the reference inherits.

It runs ONCE per sampling call (amortised over the 100 diffusion steps), so it is plain PyTorch-ROCm
ops on the device (MIOpen convolutions), evaluation-mode BatchNorm folded into scale/shift; it is
not a kernel target of this build (SURVEY.md section 8 row A0, and (f)-2 for a later round).

PARITY UNPINNED: torchvision is absent from the build container, so the reference's encoder
cannot be instantiated to emit golden vectors; the tests check this module
against an independent nn.Module assembly of the same layers, CPU vs GPU.
The state_dict keys are the reference's (`map_encoder.encoder_heads.map_model.*`,
`map_encoder.decoder.*`); the `encoder_heads.` level comes from torch.fx's feature extractor.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

PREFIX = "map_encoder."
_TRUNK = "map_encoder.encoder_heads.map_model."
_DEC = "map_encoder.decoder."
_BN_EPS = 1e-5
_LN_EPS = 1e-5


def map_encoder_shapes(in_channels=2, grid_feature_dim=32, image_hw=(224, 224)):
    """name -> shape of every tensor of the reference's MapEncoder(resnet18, grid features only)."""
    s = OrderedDict()

    def bn(prefix, c):
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)
        s[prefix + ".running_mean"] = (c,)
        s[prefix + ".running_var"] = (c,)

    s[_TRUNK + "conv1.weight"] = (64, in_channels, 7, 7)
    bn(_TRUNK + "bn1", 64)
    cin = 64
    for li, c in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = "%slayer%d.%d." % (_TRUNK, li, bi)
            s[p + "conv1.weight"] = (c, cin if bi == 0 else c, 3, 3)
            bn(p + "bn1", c)
            s[p + "conv2.weight"] = (c, c, 3, 3)
            bn(p + "bn2", c)
            if bi == 0 and li > 1:
                s[p + "downsample.0.weight"] = (c, cin, 1, 1)
                bn(p + "downsample.1", c)
        cin = c
    s[_DEC + "conv1.0.weight"] = (512, 512, 3, 3)
    for name, ci, co in (("up1", 512 + 256, 256), ("up2", 256 + 128, 128), ("up3", 128 + 64, 64)):
        p = "%s%s.conv.double_conv." % (_DEC, name)
        mid = ci // 2
        s[p + "0.weight"] = (mid, ci, 3, 3)
        s[p + "0.bias"] = (mid,)
        bn(p + "1", mid)
        s[p + "3.weight"] = (co, mid, 3, 3)
        s[p + "3.bias"] = (co,)
        bn(p + "4", co)
    p = _DEC + "conv2."
    s[p + "conv1.weight"] = (64, 64, 1, 1)
    bn(p + "bn1", 64)
    s[p + "conv2.weight"] = (64, 64, 3, 3)
    bn(p + "bn2", 64)
    s[p + "conv3.weight"] = (64, 64, 1, 1)
    bn(p + "bn3", 64)
    s[p + "shortcut.0.weight"] = (64, 64, 1, 1)
    bn(p + "shortcut.1", 64)
    s[_DEC + "conv3.weight"] = (grid_feature_dim, 64, 1, 1)
    s[_DEC + "conv3.bias"] = (grid_feature_dim,)
    s[_DEC + "out_norm.weight"] = (grid_feature_dim, image_hw[0] // 4, image_hw[1] // 4)
    s[_DEC + "out_norm.bias"] = (grid_feature_dim, image_hw[0] // 4, image_hw[1] // 4)
    return s


def random_map_encoder_state_dict(in_channels=2, grid_feature_dim=32, image_hw=(224, 224), seed=0):
    """Random-init weights of the reference's shapes (He-normal convs, BN near identity with
    non-trivial running statistics so that the folding is exercised)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in map_encoder_shapes(in_channels, grid_feature_dim, image_hw).items():
        if k.endswith("running_var"):
            t = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            t = 0.1 * torch.randn(shp, generator=g)
        elif "out_norm" in k or (len(shp) == 1 and k.endswith("weight")):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            t = 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            t = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
        sd[k] = t
    return sd


def has_map_encoder(state_dict):
    return any(k.startswith(PREFIX) and k.endswith("map_model.conv1.weight") for k in state_dict)


class MapEncoder:
    """callable(image (B,C,H,W)) -> (None, map_grid_feat (B, grid_dim, H/4, W/4)), the return
    convention of the reference's MapEncoder.forward (trace_helpers.py:219-229) without a global
    feature head (TRACE configs use the grid only: use_map_feat_global=False)."""

    # From this many images per call the activations and weights go channels-last (NHWC): MIOpen then takes
    # its NHWC implicit-GEMM fp32 kernels, 4.7x faster in steady state on MI355X (1024 images: 104 ms against
    # 489 ms, scripts/prof_map_encoder.py; max relative difference 3.9e-6) -- at the price of ~90 s of kernel compilation
    # the first time a process sees the shapes (MIOpen caches them on disk), which small calls do not repay.
    CHANNELS_LAST_FROM = 1024

    def __init__(self, state_dict, device="cuda:0", channels_last_from=None):
        self.device = torch.device(device)
        self.channels_last_from = self.CHANNELS_LAST_FROM if channels_last_from is None else channels_last_from
        self._p_nhwc = None
        p = {}
        for k, v in state_dict.items():
            if not k.startswith(PREFIX) or k.endswith("num_batches_tracked"):
                continue
            # tolerate a state_dict saved without the fx wrapper level
            kk = k.replace("map_encoder.map_model.", _TRUNK)
            p[kk] = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
        missing = [k for k in map_encoder_shapes(p[_TRUNK + "conv1.weight"].shape[1],
                                                 p[_DEC + "conv3.weight"].shape[0],
                                                 tuple(4 * d for d in p[_DEC + "out_norm.weight"].shape[1:]))
                   if k not in p]
        if missing:
            raise KeyError("map encoder weights missing from the state_dict: %s ..." % missing[:4])
        self.p = self._p_nchw = p
        self._bn = {}

    # evaluation-mode BatchNorm as a per-channel affine (folded once)
    def _bn_affine(self, prefix):
        t = self._bn.get(prefix)
        if t is None:
            p = self.p
            scale = p[prefix + ".weight"] / torch.sqrt(p[prefix + ".running_var"] + _BN_EPS)
            shift = p[prefix + ".bias"] - p[prefix + ".running_mean"] * scale
            t = (scale.view(1, -1, 1, 1), shift.view(1, -1, 1, 1))
            self._bn[prefix] = t
        return t

    def _bn_apply(self, x, prefix):
        sc, sh = self._bn_affine(prefix)
        return x * sc + sh

    def _basic_block(self, x, prefix, stride):
        p = self.p
        out = F.conv2d(x, p[prefix + "conv1.weight"], None, stride, 1)
        out = F.relu(self._bn_apply(out, prefix + "bn1"))
        out = F.conv2d(out, p[prefix + "conv2.weight"], None, 1, 1)
        out = self._bn_apply(out, prefix + "bn2")
        if prefix + "downsample.0.weight" in p:
            x = self._bn_apply(F.conv2d(x, p[prefix + "downsample.0.weight"], None, stride, 0),
                               prefix + "downsample.1")
        return F.relu(out + x)

    def _up(self, x1, x2, name):
        """base_models.py:190-227 (bilinear, align_corners=True)."""
        p = self.p
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        x = torch.cat([x2, x1], dim=1)
        q = "%s%s.conv.double_conv." % (_DEC, name)
        x = F.relu(self._bn_apply(F.conv2d(x, p[q + "0.weight"], p[q + "0.bias"], 1, 1), q + "1"))
        return F.relu(self._bn_apply(F.conv2d(x, p[q + "3.weight"], p[q + "3.bias"], 1, 1), q + "4"))

    @torch.no_grad()
    def __call__(self, image):
        x = image.to(device=self.device, dtype=torch.float32)
        nhwc = self.device.type == "cuda" and x.shape[0] >= self.channels_last_from
        if nhwc:
            if self._p_nhwc is None:
                self._p_nhwc = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v)
                                for k, v in self._p_nchw.items()}
            x = x.contiguous(memory_format=torch.channels_last)
        # the helper methods read self.p: point it at the layout of this call
        self.p = self._p_nhwc if nhwc else self._p_nchw
        try:
            return self._forward(x)
        finally:
            self.p = self._p_nchw

    def _forward(self, x):
        p = self.p
        x = F.conv2d(x, p[_TRUNK + "conv1.weight"], None, 2, 3)
        x = F.relu(self._bn_apply(x, _TRUNK + "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
        feats = []
        for li in range(1, 5):
            for bi in range(2):
                x = self._basic_block(x, "%slayer%d.%d." % (_TRUNK, li, bi), 2 if (bi == 0 and li > 1) else 1)
            feats.append(x)
        # MapGridDecoder.forward, trace_helpers.py:259-268
        x = F.relu(F.conv2d(feats[3], p[_DEC + "conv1.0.weight"], None, 1, 1))
        x = self._up(x, feats[2], "up1")
        x = self._up(x, feats[1], "up2")
        x = self._up(x, feats[0], "up3")
        q = _DEC + "conv2."                                    # ConvBlock, base_models.py:229-260
        out = F.relu(self._bn_apply(F.conv2d(x, p[q + "conv1.weight"]), q + "bn1"))
        out = F.relu(self._bn_apply(F.conv2d(out, p[q + "conv2.weight"], None, 1, 1), q + "bn2"))
        out = self._bn_apply(F.conv2d(out, p[q + "conv3.weight"]), q + "bn3")
        out = out + self._bn_apply(F.conv2d(x, p[q + "shortcut.0.weight"]), q + "shortcut.1")
        x = F.relu(out)
        x = F.conv2d(x, p[_DEC + "conv3.weight"], p[_DEC + "conv3.bias"])
        w, b = p[_DEC + "out_norm.weight"], p[_DEC + "out_norm.bias"]
        x = F.layer_norm(x.contiguous(), tuple(w.shape), w, b, _LN_EPS)
        return None, x
