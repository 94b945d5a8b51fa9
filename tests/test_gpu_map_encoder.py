"""Map encoder kernels (trace_amd/csrc/conv2d.hip) through the C ABI against fp32 / fp64
torch on the CPU: every built trace_conv2d shape with the fused scale / shift / residual / ReLU epilogue and channel
windows, the stem, max-pool, bilinear up-sampling and LayerNorm kernels, and the assembled encoder against the
functional torch path and the independent nn.Module assembly of tests/test_map_encoder_cpu.py.

PARITY: the decoder half is pinned against the reference's own MapGridDecoder (tests/golden/map_decoder.npz,
`test_hip_decoder_half_matches_the_reference_map_grid_decoder`); the ResNet-18 trunk is UNPINNED against the reference
(torchvision absent): the checker there is torch's fp32 CPU implementation of the same layers.  Tolerances are fp32 round-off of K-long dot products:
|got - want| <= 2e-5 * max|want| per layer, 3e-4 * max|want| after the 30 layers of the encoder."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib():
    from trace_amd import _cabi
    return _cabi, _cabi.lib()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# (k, stride, input H = W, cin, cout): every (filter, stride, output width) instantiation, with the channel counts
# of the network including the ragged 96-channel column group and the 32-channel head
CONV_SHAPES = [
    (3, 1, 56, 64, 64), (3, 1, 56, 192, 96), (3, 1, 56, 96, 64), (3, 1, 28, 128, 128), (3, 1, 28, 384, 192),
    (3, 1, 14, 256, 256), (3, 1, 7, 512, 512), (3, 2, 56, 64, 128), (3, 2, 28, 128, 256), (3, 2, 14, 256, 512),
    (1, 1, 56, 64, 32), (1, 2, 56, 64, 128), (1, 2, 28, 128, 256), (1, 2, 14, 256, 512),
]


@pytest.mark.parametrize("k,stride,hw,cin,cout", CONV_SHAPES)
def test_conv2d_matches_torch(k, stride, hw, cin, cout):
    from trace_amd.map_encoder import pack_conv2d_weight
    cabi, lib = _lib()
    g = torch.Generator().manual_seed(hw * 7 + cin + k)
    B = 3
    in_ctot, in_coff = cin + 32, 16            # input is a channel window of a wider buffer
    x = torch.randn(B, in_ctot, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    scale = 1.0 + 0.2 * torch.randn(cout, generator=g)
    shift = 0.3 * torch.randn(cout, generator=g)
    ho = (hw + 2 * (k // 2) - k) // stride + 1
    res_ctot, res_coff = cout + 16, 8
    res = torch.randn(B, res_ctot, ho, ho, generator=g)
    out_ctot, out_coff = cout + 48, 32
    want = F.conv2d(x[:, in_coff:in_coff + cin].double(), w.double(), None, stride, k // 2)
    want = want * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = F.relu(want + res[:, res_coff:res_coff + cout].double()).float()

    xd, rd = _nhwc(x).to(DEV), _nhwc(res).to(DEV)
    out = torch.full((B, ho, ho, out_ctot), -7.0, device=DEV)
    wp = pack_conv2d_weight(w).to(DEV)
    sd, hd = scale.to(DEV), shift.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    cabi.check(lib.trace_conv2d(xd.data_ptr(), B, hw, hw, in_ctot, in_coff, cin, wp.data_ptr(), k, k, stride, k // 2,
                                out.data_ptr(), out_ctot, out_coff, cout, sd.data_ptr(), hd.data_ptr(),
                                rd.data_ptr(), res_ctot, res_coff, 1, st), "trace_conv2d")
    torch.cuda.synchronize()
    got = out.cpu()
    # channels outside the output window are untouched
    assert bool((got[..., :out_coff] == -7.0).all()) and bool((got[..., out_coff + cout:] == -7.0).all())
    got = got[..., out_coff:out_coff + cout].permute(0, 3, 1, 2)
    err = float((got - want).abs().max())
    assert err <= 2e-5 * float(want.abs().max()), err

    # no scale / shift / residual / ReLU: the plain convolution
    out2 = torch.empty(B, ho, ho, cout, device=DEV)
    cabi.check(lib.trace_conv2d(xd.data_ptr(), B, hw, hw, in_ctot, in_coff, cin, wp.data_ptr(), k, k, stride, k // 2,
                                out2.data_ptr(), cout, 0, cout, None, None, None, 0, 0, 0, st), "trace_conv2d")
    want2 = F.conv2d(x[:, in_coff:in_coff + cin].double(), w.double(), None, stride, k // 2).float()
    err = float((out2.cpu().permute(0, 3, 1, 2) - want2).abs().max())
    assert err <= 2e-5 * float(want2.abs().max()), err


def test_conv2d_rejects_unbuilt_shapes_and_bad_windows():
    from trace_amd.map_encoder import pack_conv2d_weight
    cabi, lib = _lib()
    x = torch.zeros(1, 40, 40, 32, device=DEV)
    w = pack_conv2d_weight(torch.zeros(16, 32, 3, 3)).to(DEV)
    out = torch.zeros(1, 40, 40, 16, device=DEV)
    args = lambda hw, cin, ctot: (x.data_ptr(), 1, hw, hw, ctot, 0, cin, w.data_ptr(), 3, 3, 1, 1, out.data_ptr(), 16,
                                  0, 16, None, None, None, 0, 0, 0, None)
    assert lib.trace_conv2d(*args(40, 32, 32)) != 0 and b"no kernel" in lib.trace_last_error()
    assert lib.trace_conv2d(*args(56, 48, 48)) != 0           # cin not a multiple of 32
    assert lib.trace_conv2d(*args(56, 32, 16)) != 0           # window outside the buffer
    # an empty batch is a no-op
    a = list(args(56, 32, 32))
    a[1] = 0
    assert lib.trace_conv2d(*a) == 0


@pytest.mark.parametrize("C", [2, 7])
def test_stem_and_maxpool_match_torch(C):
    cabi, lib = _lib()
    g = torch.Generator().manual_seed(C)
    B, H = 3, 224
    img = (torch.rand(B, C, H, H, generator=g) > 0.5).float() + 0.1 * torch.randn(B, C, H, H, generator=g)
    w = torch.randn(64, C, 7, 7, generator=g) * 0.1
    scale, shift = 1.0 + 0.2 * torch.randn(64, generator=g), 0.2 * torch.randn(64, generator=g)
    want = F.relu(F.conv2d(img.double(), w.double(), None, 2, 3) * scale.double().view(1, -1, 1, 1)
                  + shift.double().view(1, -1, 1, 1)).float()
    st = torch.cuda.current_stream().cuda_stream
    s0 = torch.empty(B, 112, 112, 64, device=DEV)
    # device operands held in names: a temporary's block would be handed to the next allocation
    imd, wd = img.to(DEV), w.permute(2, 3, 1, 0).contiguous().to(DEV)
    sd, hd = scale.to(DEV), shift.to(DEV)
    cabi.check(lib.trace_map_stem(imd.data_ptr(), wd.data_ptr(), sd.data_ptr(), hd.data_ptr(), s0.data_ptr(),
                                  B, C, H, H, st), "trace_map_stem")
    got = s0.cpu().permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    pooled = torch.empty(B, 56, 56, 64, device=DEV)
    cabi.check(lib.trace_maxpool3x3s2(s0.data_ptr(), pooled.data_ptr(), B, 112, 112, 64, st), "trace_maxpool3x3s2")
    assert torch.equal(pooled.cpu().permute(0, 3, 1, 2), F.max_pool2d(got, 3, 2, 1))


@pytest.mark.parametrize("hw,C", [(7, 512), (14, 256), (28, 128)])
def test_upsample_matches_torch(hw, C):
    cabi, lib = _lib()
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(2, C, hw, hw, generator=g)
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    ctot, coff = C + 64, 64
    out = torch.full((2, 2 * hw, 2 * hw, ctot), 5.0, device=DEV)
    xd = _nhwc(x).to(DEV)
    cabi.check(lib.trace_upsample2x(xd.data_ptr(), out.data_ptr(), 2, hw, hw, C, ctot, coff,
                                    torch.cuda.current_stream().cuda_stream), "trace_upsample2x")
    got = out.cpu()
    assert bool((got[..., :coff] == 5.0).all())
    assert float((got[..., coff:].permute(0, 3, 1, 2) - want).abs().max()) <= 2e-6 * float(want.abs().max())


def test_layernorm_matches_torch():
    cabi, lib = _lib()
    g = torch.Generator().manual_seed(0)
    B, C, H = 3, 32, 56
    x = 3.0 + 2.0 * torch.randn(B, C, H, H, generator=g)
    w, b = 1.0 + 0.1 * torch.randn(C, H, H, generator=g), 0.1 * torch.randn(C, H, H, generator=g)
    want = F.layer_norm(x.double(), (C, H, H), w.double(), b.double(), 1e-5).float()
    out = torch.empty(B, H, H, C, device=DEV)
    xd = _nhwc(x).to(DEV)
    wd, bd = w.permute(1, 2, 0).contiguous().to(DEV), b.permute(1, 2, 0).contiguous().to(DEV)
    cabi.check(lib.trace_layernorm_image(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), B,
                                         C * H * H, 1e-5, torch.cuda.current_stream().cuda_stream),
               "trace_layernorm_image")
    assert float((out.cpu().permute(0, 3, 1, 2) - want).abs().max()) <= 5e-6 * float(want.abs().max())


@pytest.mark.parametrize("cin,B", [(2, 5), (7, 2)])
def test_map_encoder_hip_matches_torch_cpu_and_module_assembly(cin, B):
    from test_map_encoder_cpu import _Ref
    from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict
    sd = random_map_encoder_state_dict(cin, 32, (224, 224), seed=5 + cin)
    g = torch.Generator().manual_seed(cin)
    img = (torch.rand(B, cin, 224, 224, generator=g) > 0.6).float()
    enc = MapEncoder(sd, DEV)
    assert enc.backend == "hip"
    enc.HIP_CHUNK = 2                              # several passes incl. a ragged last one
    glob, got = enc(img)
    assert glob is None and got.shape == (B, 32, 56, 56)
    got = got.cpu()
    _, want = MapEncoder(sd, "cpu")(img)
    ref = _Ref(cin, 32, 56).eval()
    ref.load_state_dict({k[len("map_encoder."):]: v for k, v in sd.items()}, strict=False)
    with torch.no_grad():
        want2 = ref(img)
    tol = 3e-4 * float(want.abs().max())
    assert float((got - want).abs().max()) <= tol, float((got - want).abs().max())
    assert float((got - want2).abs().max()) <= tol
    # one pass over the whole batch gives the same bits as the chunked passes (no cross-image state)
    enc2 = MapEncoder(sd, DEV)
    _, got_whole = enc2(img)
    assert torch.equal(got_whole.cpu(), got)
    # the unconditional image (constant fill) encodes identically for every agent
    _, g2 = enc2(torch.ones_like(img) * 0.5)
    assert float((g2[0] - g2[1]).abs().max()) == 0.0


def test_hip_decoder_half_matches_the_reference_map_grid_decoder():
    """`MapEncoder.decode` on the library's kernels (conv2d / upsample / LayerNorm through the C ABI) against the
    output the REFERENCE's MapGridDecoder gave for the same weights and features
    (/root/reference/tbsim/models/trace_helpers.py:231-268, recorded by oracle/gen_golden_map_decoder.py), at the
    tolerance the assembled-encoder test uses (3e-4 of the output's scale: 14 convs + LayerNorm of split-bf16 fp32)."""
    from test_map_encoder_cpu import _decoder_fixture          # tests/ is on sys.path (pytest's rootdir mode)
    from trace_amd.map_encoder import MapEncoder
    sd, feats, want = _decoder_fixture()
    enc = MapEncoder(sd, device=DEV, backend="hip")
    got = enc.decode([f.to(DEV) for f in feats]).cpu()
    assert got.shape == want.shape
    err = float((got - want).abs().max())
    assert err <= 3e-4 * float(want.abs().max()), err
    # bit-reproducible, and independent of what the buffers held before
    enc(torch.randn(2, 2, 224, 224, device=DEV))
    assert torch.equal(enc.decode([f.to(DEV) for f in feats]).cpu(), got)


def test_sampler_accepts_the_channels_last_grid_of_the_hip_encoder():
    """The encoder returns (B, C, H, W) on channels-last memory; the sampler's grid copy must take it as is."""
    from trace_amd.map_encoder import MapEncoder, random_map_encoder_state_dict
    sd = random_map_encoder_state_dict(2, 32, (224, 224), seed=1)
    img = (torch.rand(3, 2, 224, 224, generator=torch.Generator().manual_seed(0)) > 0.6).float()
    _, grid = MapEncoder(sd, DEV)(img)
    _, grid_t = MapEncoder(sd, DEV, backend="torch")(img)
    assert grid.permute(0, 2, 3, 1).is_contiguous()
    assert float((grid - grid_t).abs().max()) <= 3e-4 * float(grid_t.abs().max())
