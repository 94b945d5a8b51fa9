"""The C-ABI library loads and exports exactly what include/trace_hip.h declares (CPU only: no
compute entry point is called)."""
import ctypes as C
import os
import re

import pytest

from trace_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "trace_hip.h")


def _header_functions():
    """name -> number of parameters, parsed from the header text."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments
    # declarations of diagnostic builds only (-DTRACE_CONV_STAMPS): not part of the shipped library
    text = re.sub(r"#ifdef TRACE_CONV_STAMPS.*?#endif", "", text, flags=re.S)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(trace_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        out[m.group(1)] = n
    return out


def test_header_and_binding_list_the_same_symbols():
    hdr = _header_functions()
    assert len(hdr) >= 20
    assert set(hdr) == set(_cabi.PROTOTYPES), (set(hdr) ^ set(_cabi.PROTOTYPES))
    for name, n in hdr.items():
        assert len(_cabi.PROTOTYPES[name][1]) == n, "%s: header has %d parameters, binding %d" % (
            name, n, len(_cabi.PROTOTYPES[name][1]))


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(_cabi.LIB_PATH), "build with `python -m trace_amd.build`"
    l = _cabi.lib()
    for name in _header_functions():
        assert hasattr(l, name), name
    assert l.trace_abi_version() == 1
    # parameter count of the orca_mixed denoiser: a host-only entry point
    from trace_amd.config import ModelConfig
    from trace_amd.unet_params import flatten_unet_params
    from trace_amd.weights import random_state_dict
    cfg = ModelConfig()
    uc = _cabi.UnetConfig(horizon=cfg.horizon, in_channels=cfg.unet_in_dim, n_levels=3,
                          dims=(C.c_int * 4)(*cfg.unet_dims[1:4], 0), cond_dim=cfg.cond_feat_dim,
                          time_dim=cfg.time_dim, n_groups=cfg.n_groups, n_steps=cfg.n_diffusion_steps)
    n = l.trace_unet_param_count(C.byref(uc))
    assert n == flatten_unet_params(cfg, random_state_dict(cfg, 0)).numel()
    # structural parameter count of the temporal UNet incl. time MLP (17.45 MB fp32)
    assert 4.2e6 < n < 4.5e6


def test_every_entry_point_cites_the_reference():
    text = open(HEADER).read()
    assert text.count("tbsim/") >= 20


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_cabi, "_lib", None)
    monkeypatch.setattr(_cabi, "LIB_PATH", str(tmp_path / "libtrace_hip.so"))
    with pytest.raises(_cabi.TraceHipError, match="no CPU fallback"):
        _cabi.lib()


def test_sampler_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from trace_amd.config import ModelConfig
    from trace_amd.sampler import TraceSampler
    from trace_amd.weights import random_state_dict
    cfg = ModelConfig()
    with pytest.raises(_cabi.TraceHipError, match="no CPU fallback"):
        TraceSampler(cfg, random_state_dict(cfg, 0), "cuda:0")
