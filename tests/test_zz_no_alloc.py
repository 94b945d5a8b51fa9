"""The entry point that allocates nothing (trace_guide_agent_collision_static, include/trace_hip.h)
against the existing trace_guide_agent_collision, through the C ABI: bit-identical loss and gradient
on two calls in a row, the second with a smaller input (the static scratch keeps the first call's
contents), plus the size limit.  Collected last on purpose (file name)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
T = 52


def _inputs(n_scenes, per_scene, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    B = n_scenes * per_scene
    M = B * N
    # agents of a scene within a few metres of each other so that pairs are inside the penalty distance
    traj = torch.zeros(M, T, 6)
    traj[..., 0:2] = torch.randn(M, T, 2, generator=g) * 0.6
    traj[..., 3] = torch.randn(M, T, generator=g)
    yaw = torch.rand(B, generator=g) * 6.28
    wfa = torch.zeros(B, 3, 3)
    wfa[:, 0, 0], wfa[:, 0, 1] = torch.cos(yaw), -torch.sin(yaw)
    wfa[:, 1, 0], wfa[:, 1, 1] = torch.sin(yaw), torch.cos(yaw)
    wfa[:, :2, 2] = torch.randn(B, 2, generator=g) * 0.8
    wfa[:, 2, 2] = 1.0
    radius = 0.3 + 0.2 * torch.rand(B, generator=g)
    disk_x = torch.linspace(-0.4, 0.4, D).repeat(B, 1) if D > 1 else torch.zeros(B, 1)
    start = (torch.arange(B) // per_scene * per_scene).int()
    size = torch.full((B,), per_scene, dtype=torch.int32)
    in_mask = (torch.rand(B, generator=g) > 0.2).int()
    scale = torch.where(in_mask > 0, torch.full((B,), 1000.0 / (per_scene * N)), torch.full((B,), -1.0))
    dev = lambda t: t.contiguous().to(DEV)
    return dict(traj=dev(traj), wfa=dev(wfa), radius=dev(radius), disk_x=dev(disk_x), start=dev(start),
                size=dev(size), in_mask=dev(in_mask), scale=dev(scale), B=B, N=N, D=D, M=M)


def _call(L, cabi, d, static):
    grad = torch.zeros(d["M"], T, 6, device=DEV)
    loss = torch.full((d["M"],), float("nan"), device=DEV)
    p = lambda k: d[k].data_ptr()
    common = (p("traj"), grad.data_ptr(), loss.data_ptr(), p("wfa"), p("radius"), p("disk_x"), d["D"], 0.2,
              p("start"), p("size"), p("in_mask"), p("scale"), d["B"], d["N"], T)
    if static:
        cabi.check(L.trace_guide_agent_collision_static(*common, None), "agent_collision_static")
    else:
        ws = torch.empty(d["M"] * T * 4, device=DEV)
        cabi.check(L.trace_guide_agent_collision(*common, ws.data_ptr(), None), "agent_collision")
    torch.cuda.synchronize()
    return loss.cpu(), grad.cpu()


def test_static_scratch_entry_point_matches_workspace_entry_point():
    from trace_amd import _cabi as cabi
    L = cabi.lib()
    cabi.check(L.trace_static_scratch_init(), "trace_static_scratch_init")
    big = _inputs(3, 24, 20, 1, seed=1)       # 1440 trajectories: 300k scratch floats
    small = _inputs(2, 5, 3, 3, seed=2)       # 30 trajectories, 3 disks: runs on the first call's leftovers
    for d in (big, small):
        want_l, want_g = _call(L, cabi, d, static=False)
        got_l, got_g = _call(L, cabi, d, static=True)
        assert float(want_g.abs().max()) > 0 and bool(torch.isfinite(want_g).all())
        assert torch.equal(torch.isnan(got_l), torch.isnan(want_l))
        assert torch.equal(torch.nan_to_num(got_l), torch.nan_to_num(want_l))
        assert torch.equal(got_g, want_g)


def test_static_scratch_entry_point_rejects_oversize():
    from trace_amd import _cabi as cabi
    L = cabi.lib()
    cap = L.trace_static_scratch_bytes() // 4
    n_agents = cap // (20 * T * 4) + 1
    d = _inputs(1, 2, 1, 1, seed=3)
    p = lambda k: d[k].data_ptr()
    rc = L.trace_guide_agent_collision_static(p("traj"), None, p("traj"), p("wfa"), p("radius"), p("disk_x"), 1, 0.2,
                                              p("start"), p("size"), p("in_mask"), p("scale"), n_agents, 20, T, None)
    assert rc == -1 and b"static scratch" in L.trace_last_error()
