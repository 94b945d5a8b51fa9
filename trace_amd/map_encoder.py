"""Map encoder of TRACE's conditioning (row A0 of SURVEY.md section 8): rasterised map image
(B, C, 224, 224) -> `map_grid_feat` (B, 32, 56, 56).

Structure restated from the reference
  * MapEncoder / MapGridDecoder      /root/reference/tbsim/models/trace_helpers.py:176-268
  * RasterizedMapEncoder (ResNet-18 trunk with a C-channel 7x7 stem)
                                      /root/reference/tbsim/models/base_models.py:402-491
  * Up / DoubleConv / ConvBlock       /root/reference/tbsim/models/base_models.py:169-260
and from torchvision's ResNet-18 (BasicBlock x [2,2,2,2]), whose layout the reference inherits.

It runs ONCE per sampling call (amortised over the 100 diffusion steps).  On the GPU, for the
224 x 224 rasters of every TRACE config, it runs on this library's own kernels (backend "hip",
trace_amd/csrc/conv2d.hip: NHWC implicit-GEMM convolutions on the bf16 matrix pipe with fp32
results, BatchNorm / bias / residual / ReLU fused, skip concatenations written in place; stem,
max-pool, bilinear up-sampling and LayerNorm kernels) through the C ABI (include/trace_hip.h,
SURVEY.md section 8(f)-2).  Backend "torch" is the same network as PyTorch-ROCm ops (MIOpen
convolutions): the comparison path of the tests, and the path for other raster sizes and the CPU.

PARITY: the decoder half (MapGridDecoder + Up / DoubleConv / ConvBlock + LayerNorm) is pinned against the
reference's own classes, which are pure torch and import through oracle/ref_harness.py
(oracle/gen_golden_map_decoder.py -> tests/golden/map_decoder.npz; `MapEncoder.decode`).  The ResNet-18 TRUNK
is torchvision's; torchvision is absent from the build container, so the reference's trunk cannot be
instantiated to emit golden vectors and that half stays unpinned -- the tests check it
against an independent nn.Module assembly of the same layers, CPU vs GPU.
The state_dict keys are the reference's (`map_encoder.encoder_heads.map_model.*`,
`map_encoder.decoder.*`); the `encoder_heads.` level comes from torch.fx's feature extractor.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import _cabi

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


def pack_conv2d_weight(w):
    """Conv2d weight (cout, cin, kh, kw) fp32 -> the operand stream of trace_conv2d: every value split
    exactly into three bf16 (w = h + m + l), laid out [cout/16][cin/32][kh*kw][plane][lane = 16 kq + col][8]
    (MFMA B fragment of v_mfma_f32_16x16x32_bf16: lane holds input channels 8 kq .. 8 kq + 7 of output
    channel col)."""
    co, ci, kh, kw = w.shape
    assert co % 16 == 0 and ci % 32 == 0, (co, ci)
    x = w.detach().to(torch.float32).cpu().reshape(co // 16, 16, ci // 32, 4, 8, kh * kw)
    x = x.permute(0, 2, 5, 3, 1, 4).contiguous()           # [col tile][chunk][tap][kq][col][j]
    h = x.bfloat16()
    r = x - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return torch.stack([h, m, l], dim=3).contiguous()


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
    # images per pass of the HIP backend: bounds the activation buffers (about 45 MB per image) and keeps a
    # layer's producer -> consumer distance inside what L2 / MALL still hold
    HIP_CHUNK = 256

    def __init__(self, state_dict, device="cuda:0", channels_last_from=None, backend=None):
        self.device = torch.device(device)
        self.backend = backend or ("hip" if self.device.type == "cuda" else "torch")
        assert self.backend in ("hip", "torch")
        if self.backend == "hip":
            _cabi.lib()                    # fails loudly when the library is not built
        self._hip = None
        self._hip_bufs = {}
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
        if self.backend == "hip":
            # no silent second path: the kernels are built for TRACE's rasters (224 x 224, 2 ORCA / 7 nuScenes
            # channels); anything else has to ask for the PyTorch form by name
            if tuple(x.shape[2:]) != (224, 224) or x.shape[1] > 8:
                raise ValueError("map encoder (hip backend): rasters of shape %s are not built (224 x 224, at most 8 "
                                 "channels); construct MapEncoder(..., backend='torch') for other shapes"
                                 % (tuple(x.shape[1:]),))
            return None, self._forward_hip(x.contiguous())
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
        return None, self._decode(feats)

    def _decode(self, feats):
        """MapGridDecoder.forward, trace_helpers.py:259-268, on the trunk's layer1..layer4 features."""
        p = self.p
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
        return F.layer_norm(x.contiguous(), tuple(w.shape), w, b, _LN_EPS)

    @torch.no_grad()
    def decode(self, feats):
        """The decoder half alone: ResNet-18 layer1..layer4 features ((B,64,56,56), (B,128,28,28), (B,256,14,14),
        (B,512,7,7)) -> map_grid_feat (B, grid_dim, 56, 56), i.e. the reference's
        `MapEncoder.forward(None, encoder_feats=...)` / `MapGridDecoder.forward` (trace_helpers.py:219-229,259-268).
        This half is pinned against the reference's own classes (tests/golden/map_decoder.npz); the trunk above it
        is torchvision's and stays unpinned."""
        feats = [f.to(device=self.device, dtype=torch.float32) for f in feats]
        assert len(feats) == 4 and [f.shape[1] for f in feats] == [64, 128, 256, 512], [f.shape for f in feats]
        if self.backend == "hip":
            if tuple(feats[0].shape[2:]) != (56, 56):
                raise ValueError("map decoder (hip backend): layer1 features of %s are not built (56 x 56); "
                                 "construct MapEncoder(..., backend='torch') for other shapes" % (tuple(feats[0].shape),))
            return self._decode_hip(feats)
        self.p = self._p_nchw
        return self._decode(feats)

    # ------------------------------------------------------------------ HIP backend
    def _prepare_hip(self):
        """Folds BatchNorm (+ conv bias) into per-channel scale / shift and packs the weights, once."""
        p, dev = self._p_nchw, self.device

        def fold(bn_prefix, bias=None):
            scale = p[bn_prefix + ".weight"] / torch.sqrt(p[bn_prefix + ".running_var"] + _BN_EPS)
            shift = p[bn_prefix + ".bias"] - p[bn_prefix + ".running_mean"] * scale
            if bias is not None:
                shift = shift + scale * bias
            return scale.contiguous(), shift.contiguous()

        def conv(wkey, bn_prefix=None, bias_key=None):
            w = p[wkey]
            bias = p[bias_key] if bias_key else None
            if bn_prefix:
                scale, shift = fold(bn_prefix, bias)
            else:
                scale, shift = None, (bias.contiguous() if bias is not None else None)
            return {"w": pack_conv2d_weight(w).to(dev), "k": w.shape[2], "cin": w.shape[1], "cout": w.shape[0],
                    "scale": scale, "shift": shift}

        h = {"stem_w": p[_TRUNK + "conv1.weight"].permute(2, 3, 1, 0).contiguous()}
        h["stem_scale"], h["stem_shift"] = fold(_TRUNK + "bn1")
        for li in range(1, 5):
            for bi in range(2):
                q = "%slayer%d.%d." % (_TRUNK, li, bi)
                h[q + "conv1"] = conv(q + "conv1.weight", q + "bn1")
                h[q + "conv2"] = conv(q + "conv2.weight", q + "bn2")
                if q + "downsample.0.weight" in p:
                    h[q + "ds"] = conv(q + "downsample.0.weight", q + "downsample.1")
        h["dec1"] = conv(_DEC + "conv1.0.weight")
        for name in ("up1", "up2", "up3"):
            q = "%s%s.conv.double_conv." % (_DEC, name)
            h[name + "a"] = conv(q + "0.weight", q + "1", q + "0.bias")
            h[name + "b"] = conv(q + "3.weight", q + "4", q + "3.bias")
        q = _DEC + "conv2."
        h["cb1"] = conv(q + "conv1.weight", q + "bn1")
        h["cb2"] = conv(q + "conv2.weight", q + "bn2")
        h["cb3"] = conv(q + "conv3.weight", q + "bn3")
        h["cbs"] = conv(q + "shortcut.0.weight", q + "shortcut.1")
        h["out"] = conv(_DEC + "conv3.weight", None, _DEC + "conv3.bias")
        # LayerNorm parameters (C, H, W) -> the NHWC memory order of the activations
        h["ln_w"] = p[_DEC + "out_norm.weight"].permute(1, 2, 0).contiguous()
        h["ln_b"] = p[_DEC + "out_norm.bias"].permute(1, 2, 0).contiguous()
        self._hip = h

    def _buffers(self, n):
        b = self._hip_bufs.get(n)
        if b is None:
            gd = self._p_nchw[_DEC + "conv3.weight"].shape[0]

            def e(hw, c):
                return torch.empty(n, hw, hw, c, device=self.device, dtype=torch.float32)

            b = {"s0": e(112, 64), "cat3": e(56, 192), "a56": e(56, 64), "b56": e(56, 64), "c56": e(56, 64),
                 "cat2": e(28, 384), "a28": e(28, 128), "b28": e(28, 128), "c28": e(28, 128),
                 "cat1": e(14, 768), "a14": e(14, 256), "b14": e(14, 256), "c14": e(14, 256),
                 "a7": e(7, 512), "b7": e(7, 512), "c7": e(7, 512), "d7": e(7, 512),
                 "m14": e(14, 384), "u1": e(14, 256), "m28": e(28, 192), "u2": e(28, 128),
                 "m56": e(56, 96), "u3": e(56, 64), "o32": e(56, gd)}
            self._hip_bufs = {n: b}          # one chunk size resident at a time
        return b

    def _conv(self, n, L, src, dst, stride=1, relu=True, res=None):
        """src / dst / res: (tensor, first channel).  One trace_conv2d launch."""
        (x, xoff), (y, yoff) = src, dst
        k = L["k"]
        r, roff = res if res is not None else (None, 0)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _cabi.check(_cabi.lib().trace_conv2d(
            x.data_ptr(), n, x.shape[1], x.shape[2], x.shape[3], xoff, L["cin"], L["w"].data_ptr(), k, k, stride,
            k // 2, y.data_ptr(), y.shape[3], yoff, L["cout"], _cabi.ptr(L["scale"]), _cabi.ptr(L["shift"]),
            _cabi.ptr(r), r.shape[3] if r is not None else 0, roff, 1 if relu else 0, st), "trace_conv2d")

    def _forward_hip(self, image):
        if self._hip is None:
            self._prepare_hip()
        h, lib = self._hip, _cabi.lib()
        B, C = image.shape[:2]
        gd = self._p_nchw[_DEC + "conv3.weight"].shape[0]
        out = torch.empty(B, 56, 56, gd, device=self.device, dtype=torch.float32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            for i0 in range(0, B, self.HIP_CHUNK):
                n = min(self.HIP_CHUNK, B - i0)
                b = self._buffers(min(self.HIP_CHUNK, B))
                cv = lambda *a, **k: self._conv(n, *a, **k)
                _cabi.check(lib.trace_map_stem(image[i0:].data_ptr(), h["stem_w"].data_ptr(),
                                               h["stem_scale"].data_ptr(), h["stem_shift"].data_ptr(),
                                               b["s0"].data_ptr(), n, C, 224, 224, st), "trace_map_stem")
                _cabi.check(lib.trace_maxpool3x3s2(b["s0"].data_ptr(), b["a56"].data_ptr(), n, 112, 112, 64, st),
                            "trace_maxpool3x3s2")
                # trunk: torchvision BasicBlock = relu(bn2(conv2(relu(bn1(conv1(x))))) + identity); the last block
                # of a layer writes its output into the skip window of the decoder's concatenation buffer
                x = (b["a56"], 0)
                for li, (ta, tb, tc, dst) in enumerate((("a56", "b56", "c56", ("cat3", 0)),
                                                        ("a28", "b28", "c28", ("cat2", 0)),
                                                        ("a14", "b14", "c14", ("cat1", 0)),
                                                        ("a7", "b7", "c7", ("d7", 0))), start=1):
                    q0, q1 = "%slayer%d.0." % (_TRUNK, li), "%slayer%d.1." % (_TRUNK, li)
                    if li == 1:
                        cv(h[q0 + "conv1"], x, (b[tb], 0))
                        cv(h[q0 + "conv2"], (b[tb], 0), (b[tc], 0), res=x)
                    else:
                        cv(h[q0 + "conv1"], x, (b[ta], 0), stride=2)
                        cv(h[q0 + "ds"], x, (b[tb], 0), stride=2, relu=False)
                        cv(h[q0 + "conv2"], (b[ta], 0), (b[tc], 0), res=(b[tb], 0))
                    t = tb if li == 1 else ta
                    cv(h[q1 + "conv1"], (b[tc], 0), (b[t], 0))
                    cv(h[q1 + "conv2"], (b[t], 0), (b[dst[0]], dst[1]), res=(b[tc], 0))
                    x = (b[dst[0]], dst[1])
                self._decode_hip_chunk(n, b, out[i0:], st)
        # (B, C, H, W) logical shape of the reference on channels-last memory: the sampler's grid layout
        return out.permute(0, 3, 1, 2)

    def _decode_hip_chunk(self, n, b, out, st):
        """MapGridDecoder.forward (trace_helpers.py:259-268) on the features the trunk left in the skip windows of
        the concatenation buffers (cat3 / cat2 / cat1, channel 0 ..) and in d7."""
        h, lib = self._hip, _cabi.lib()
        gd = self._p_nchw[_DEC + "conv3.weight"].shape[0]
        cv = lambda *a, **k: self._conv(n, *a, **k)
        cv(h["dec1"], (b["d7"], 0), (b["a7"], 0))
        src = b["a7"]
        for name, cat, skip_c, mid, o in (("up1", "cat1", 256, "m14", "u1"), ("up2", "cat2", 128, "m28", "u2"),
                                          ("up3", "cat3", 64, "m56", "u3")):
            _cabi.check(lib.trace_upsample2x(src.data_ptr(), b[cat].data_ptr(), n, src.shape[1], src.shape[2],
                                             src.shape[3], b[cat].shape[3], skip_c, st), "trace_upsample2x")
            cv(h[name + "a"], (b[cat], 0), (b[mid], 0))
            cv(h[name + "b"], (b[mid], 0), (b[o], 0))
            src = b[o]
        # ConvBlock, base_models.py:229-260
        cv(h["cb1"], (b["u3"], 0), (b["a56"], 0))
        cv(h["cb2"], (b["a56"], 0), (b["b56"], 0))
        cv(h["cbs"], (b["u3"], 0), (b["c56"], 0), relu=False)
        cv(h["cb3"], (b["b56"], 0), (b["a56"], 0), res=(b["c56"], 0))
        cv(h["out"], (b["a56"], 0), (b["o32"], 0), relu=False)
        _cabi.check(lib.trace_layernorm_image(b["o32"].data_ptr(), h["ln_w"].data_ptr(), h["ln_b"].data_ptr(),
                                              out.data_ptr(), n, 56 * 56 * gd, _LN_EPS, st),
                    "trace_layernorm_image")

    def _decode_hip(self, feats):
        if self._hip is None:
            self._prepare_hip()
        B = feats[0].shape[0]
        gd = self._p_nchw[_DEC + "conv3.weight"].shape[0]
        out = torch.empty(B, 56, 56, gd, device=self.device, dtype=torch.float32)
        st = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            for i0 in range(0, B, self.HIP_CHUNK):
                n = min(self.HIP_CHUNK, B - i0)
                b = self._buffers(min(self.HIP_CHUNK, B))
                # NCHW features -> the NHWC skip windows the trunk would have written
                for f, (name, c) in zip(feats, (("cat3", 64), ("cat2", 128), ("cat1", 256), ("d7", 512))):
                    b[name][:n, :, :, :c] = f[i0:i0 + n].permute(0, 2, 3, 1)
                self._decode_hip_chunk(n, b, out[i0:], st)
        return out.permute(0, 3, 1, 2)
