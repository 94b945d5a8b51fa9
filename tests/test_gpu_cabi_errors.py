"""Error behaviour of the C ABI (include/trace_hip.h): bad arguments return a negative status and a
message through trace_last_error(); nothing is launched, nothing faults."""
import ctypes as C

import pytest
import torch

from trace_amd import _cabi
from trace_amd.config import ModelConfig
from trace_amd.sampler import TraceSampler, C_PAD
from trace_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu
EINVAL, EHIP, ENOMEM = -1, -2, -3


@pytest.fixture(scope="module")
def sampler():
    cfg = ModelConfig()
    return TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0")


def _err(L):
    m = L.trace_last_error()
    return m.decode() if m else ""


def test_forward_rejects_small_workspace_and_bad_step(sampler):
    s, L = sampler, sampler.lib
    rows = 40
    x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0")
    table = torch.zeros(rows // 20, s.cond_width, device="cuda:0")
    pred = torch.empty(rows, 52, 2, device="cuda:0")
    need = L.trace_unet_workspace_bytes(s._unet, rows, 1)
    assert need > 0 and need >= L.trace_unet_workspace_bytes(s._unet, rows, 0)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    rc = L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 10, rows, pred.data_ptr(),
                              ws.data_ptr(), need // 2, 1, None)
    assert rc == ENOMEM and "workspace too small" in _err(L)
    rc = L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 100, rows, pred.data_ptr(),
                              ws.data_ptr(), need, 1, None)
    assert rc == EINVAL and "step" in _err(L)
    rc = L.trace_unet_forward(s._unet, None, table.data_ptr(), 20, 10, rows, pred.data_ptr(), ws.data_ptr(),
                              need, 1, None)
    assert rc == EINVAL and "null" in _err(L)
    # zero rows is a no-op, not an error
    assert L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 10, 0, pred.data_ptr(),
                                ws.data_ptr(), need, 1, None) == 0
    # and the valid call still works afterwards
    assert L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 10, rows, pred.data_ptr(),
                                ws.data_ptr(), need, 1, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()


def test_backward_rejects_small_workspace(sampler):
    s, L = sampler, sampler.lib
    rows = 40
    g = torch.zeros(rows, 52, 2, device="cuda:0")
    ws = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
    rc = L.trace_unet_backward_input(s._unet, g.data_ptr(), rows, g.data_ptr(), ws.data_ptr(), 1024, None)
    assert rc == ENOMEM and "workspace" in _err(L)


def test_create_rejects_wrong_parameter_count_and_shape():
    L = _cabi.lib()
    cfg = ModelConfig()
    uc = _cabi.UnetConfig(horizon=cfg.horizon, in_channels=cfg.unet_in_dim, n_levels=3,
                          dims=(C.c_int * 4)(*cfg.unet_dims[1:4], 0), cond_dim=cfg.cond_feat_dim,
                          time_dim=cfg.time_dim, n_groups=cfg.n_groups, n_steps=cfg.n_diffusion_steps)
    n = L.trace_unet_param_count(C.byref(uc))
    buf = torch.zeros(n)
    h = C.c_void_p()
    assert L.trace_unet_create(C.byref(uc), buf.data_ptr(), n - 1, C.byref(h)) == EINVAL
    assert "parameters" in _err(L) and not h.value
    uc.horizon = 48
    assert L.trace_unet_create(C.byref(uc), buf.data_ptr(), n, C.byref(h)) == EINVAL
    assert "horizon" in _err(L)


def test_profile_hooks_count_conv_launches(sampler):
    s, L = sampler, sampler.lib
    rows = 40
    x_in = torch.zeros(rows, C_PAD, 52, device="cuda:0")
    table = torch.zeros(rows // 20, s.cond_width, device="cuda:0")
    pred = torch.empty(rows, 52, 2, device="cuda:0")
    need = L.trace_unet_workspace_bytes(s._unet, rows, 0)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    assert L.trace_profile_begin(0) == EINVAL
    assert L.trace_profile_begin(64) == 0
    assert L.trace_unet_forward(s._unet, x_in.data_ptr(), table.data_ptr(), 20, 10, rows, pred.data_ptr(),
                                ws.data_ptr(), need, 0, None) == 0
    ms, nl, fl = C.c_double(), C.c_int64(), C.c_double()
    assert L.trace_profile_end(C.byref(ms), C.byref(nl), C.byref(fl)) == 0
    # 34 convs; at this size the k5 conv and the 1x1 residual conv of blocks 2, 4, 8, 10 share a launch
    assert nl.value == 30 and ms.value > 0
    # SURVEY.md 8d counts 60,288,512 MACs per trajectory per denoiser forward.  Of these the conv
    # kernels do all but: the 1x1 head (2*64*52, its own kernel), the 12 per-block cond/time
    # Linears (1792 * 288, tabulated once per call / at create time) and the time MLP (2*32*128).
    per_row = fl.value / rows
    assert per_row == 2 * (60288512 - 2 * 64 * 52 - 1792 * 288 - 2 * 32 * 128), per_row


def test_env_rasterize_and_social_group_reject_bad_arguments(sampler):
    L = sampler.lib
    dev = "cuda:0"
    pose = torch.zeros(2, 3, device=dev)
    soa = torch.zeros(2, dtype=torch.int32, device=dev)
    boxes = torch.zeros(1, 1, 4, device=dev)
    rfa = (C.c_float * 9)(12, 0, 56, 0, 12, 112, 0, 0, 1)
    img = torch.empty(2, 2, 8, 8, device=dev)
    drv = torch.empty(2, 8, 8, dtype=torch.uint8, device=dev)
    ok = L.trace_env_rasterize(pose.data_ptr(), soa.data_ptr(), boxes.data_ptr(), 1, C.byref(rfa), img.data_ptr(), 2,
                               drv.data_ptr(), 2, 8, 8, None)
    assert ok == 0
    torch.cuda.synchronize()
    assert bool(drv.bool().all())                      # a zero-size box blocks nothing
    rc = L.trace_env_rasterize(pose.data_ptr(), soa.data_ptr(), boxes.data_ptr(), 1, C.byref(rfa), img.data_ptr(), 2,
                               drv.data_ptr(), 2, 8, 6, None)
    assert rc == EINVAL and "multiple of 4" in _err(L)
    rc = L.trace_env_rasterize(pose.data_ptr(), soa.data_ptr(), boxes.data_ptr(), 1, C.byref(rfa), None, 2,
                               None, 2, 8, 8, None)
    assert rc == EINVAL and "null" in _err(L)
    skew = (C.c_float * 9)(12, 1, 56, 0, 12, 112, 0, 0, 1)
    rc = L.trace_env_rasterize(pose.data_ptr(), soa.data_ptr(), boxes.data_ptr(), 1, C.byref(skew), img.data_ptr(), 2,
                               drv.data_ptr(), 2, 8, 8, None)
    assert rc == EINVAL and "scale + offset" in _err(L)
    assert L.trace_env_rasterize(pose.data_ptr(), soa.data_ptr(), boxes.data_ptr(), 1, C.byref(rfa), img.data_ptr(), 2,
                                 drv.data_ptr(), 0, 8, 8, None) == 0          # no agents: nothing to do
    # social group: a members table narrower than two columns cannot hold a group
    M, T = 4, 52
    traj = torch.zeros(M, T, 6, device=dev)
    loss = torch.zeros(M, device=dev)
    wfa = torch.eye(3, device=dev).repeat(2, 1, 1).contiguous()
    members = torch.full((2, 8), -1, dtype=torch.int32, device=dev)
    lead = torch.zeros(2, dtype=torch.int32, device=dev)
    scale = torch.ones(2, device=dev)
    ws = torch.empty(M * T * 4, device=dev)
    nbr = torch.empty(M * T, dtype=torch.int32, device=dev)
    dist = torch.empty(M * T, device=dev)
    rc = L.trace_guide_social_group(traj.data_ptr(), None, loss.data_ptr(), wfa.data_ptr(), members.data_ptr(), 1,
                                    lead.data_ptr(), None, 1.5, scale.data_ptr(), 2, 2, T, ws.data_ptr(),
                                    nbr.data_ptr(), dist.data_ptr(), None)
    assert rc == EINVAL and "width" in _err(L)
    rc = L.trace_guide_social_group(traj.data_ptr(), None, loss.data_ptr(), wfa.data_ptr(), members.data_ptr(), 25,
                                    lead.data_ptr(), None, 1.5, scale.data_ptr(), 2, 2, T, None,
                                    nbr.data_ptr(), dist.data_ptr(), None)
    assert rc == EINVAL and "null" in _err(L)
