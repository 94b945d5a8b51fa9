"""ctypes binding of libtrace_hip.so (C ABI declared in include/trace_hip.h).

There is no CPU fallback: if the library is missing or a symbol cannot be resolved,
importing/using the product path raises.  Build it with ``python -m trace_amd.build``.
"""
import ctypes as C
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  It
# must be loaded BEFORE libtrace_hip.so so that both share ONE runtime instance (streams and
# device pointers cross the boundary); the other order leaves two runtimes in the process and
# the second one reports "no ROCm-capable device".
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRACE_HIP_LIB: diagnostic builds only (e.g. the time-stamped library of scripts/stamps.py)
LIB_PATH = os.environ.get("TRACE_HIP_LIB") or os.path.join(_HERE, "lib", "libtrace_hip.so")


class TraceHipError(RuntimeError):
    pass


class UnetConfig(C.Structure):
    _fields_ = [("horizon", C.c_int), ("in_channels", C.c_int), ("n_levels", C.c_int),
                ("dims", C.c_int * 4), ("cond_dim", C.c_int), ("time_dim", C.c_int),
                ("n_groups", C.c_int), ("n_steps", C.c_int)]


class DynParams(C.Structure):
    _fields_ = [("norm_add", C.c_float * 6), ("norm_div", C.c_float * 6), ("dt", C.c_float),
                ("acc_lo", C.c_float), ("acc_hi", C.c_float), ("v_lo", C.c_float),
                ("v_hi", C.c_float), ("max_steer", C.c_float), ("max_yawvel", C.c_float)]


class StepCoefs(C.Structure):
    _fields_ = [("sqrt_recip_alphas_cumprod", C.c_float),
                ("sqrt_recipm1_alphas_cumprod", C.c_float),
                ("posterior_mean_coef1", C.c_float), ("posterior_mean_coef2", C.c_float),
                ("posterior_log_variance_clipped", C.c_float),
                ("sqrt_alphas_over_one_minus_alphas_cumprod", C.c_float),
                ("sqrt_recip_one_minus_alphas_cumprod", C.c_float), ("t", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64

# name -> (restype, argtypes).  Must list every symbol include/trace_hip.h declares
# (tests/test_cabi_symbols.py checks this against the header text).
PROTOTYPES = {
    "trace_last_error": (C.c_char_p, []),
    "trace_abi_version": (_I, []),
    "trace_unet_param_count": (_L, [C.POINTER(UnetConfig)]),
    "trace_unet_create": (_I, [C.POINTER(UnetConfig), _P, _L, C.POINTER(_P)]),
    "trace_unet_destroy": (None, [_P]),
    "trace_unet_cond_width": (_I, [_P]),
    "trace_unet_cond_table": (_I, [_P, _P, _I, _P, _P]),
    "trace_unet_workspace_bytes": (_L, [_P, _I, _I]),
    "trace_unet_forward": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _L, _I, _P]),
    "trace_unet_backward_input": (_I, [_P, _P, _I, _P, _P, _L, _P]),
    "trace_profile_begin": (_I, [_I]),
    "trace_profile_end": (_I, [C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double)]),
    "trace_profile_pipe_seconds": (C.c_double, []),
    "trace_map_query": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F,
                             C.POINTER(DynParams), _P]),
    "trace_map_query_cfg": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F,
                                 C.POINTER(DynParams), _P]),
    "trace_posterior_step_cfg": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                      C.POINTER(StepCoefs), C.POINTER(DynParams), _P]),
    "trace_dyn_rollout": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(DynParams), _P]),
    "trace_dyn_adjoint": (_I, [_P, _P, _P, _P, _I, _I, _I, C.POINTER(DynParams), _P]),
    "trace_cfg_combine": (_I, [_P, _P, _P, _P, _I, _I, _F, C.POINTER(StepCoefs), _P]),
    "trace_cfg_combine_adjoint": (_I, [_P, _P, _P, _I, _I, _F, C.POINTER(StepCoefs), _P]),
    "trace_posterior_mean": (_I, [_P, _P, _P, _I, _I, C.POINTER(StepCoefs), _P]),
    "trace_posterior_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                  C.POINTER(StepCoefs), C.POINTER(DynParams), _P]),
    "trace_guide_target_pos": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "trace_guide_target_pos_at_time": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "trace_guide_progress": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "trace_guide_speed": (_I, [_P, _P, _P, _P, _P, _I, _F, _I, _I, _I, _P]),
    "trace_guide_agent_collision": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _I,
                                         _I, _P, _P]),
    "trace_static_scratch_init": (_I, []),
    "trace_static_scratch_bytes": (_L, []),
    "trace_guide_agent_collision_static": (_I, [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _I,
                                                _I, _P]),
    "trace_guide_map_collision": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I,
                                       _P, _P]),
    "trace_guide_map_collision_argmin": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I,
                                              _P, _P, _P]),
    "trace_env_rasterize": (_I, [_P, _P, _P, _I, C.POINTER(C.c_float * 9), _P, _I, _P, _I, _I, _I, _P]),
    "trace_guide_social_group": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _F, _P, _I, _I, _I, _P, _P, _P, _P]),
    "trace_conv2d": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P]),
    "trace_conv_ws_mode": (_I, [_I]),
    "trace_conv_scheme": (_I, [_I]),
    "trace_map_stem": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "trace_maxpool3x3s2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "trace_upsample2x": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "trace_layernorm_image": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
}

_lib = None


def lib():
    """Loads the shared library (once).  Raises TraceHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TraceHipError(
            "HIP library not built: %s is missing (run `python -m trace_amd.build`); "
            "there is no CPU fallback for the sampler" % LIB_PATH)
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise TraceHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            raise TraceHipError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc, what=""):
    if rc != 0:
        msg = lib().trace_last_error()
        raise TraceHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device/host pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor passed to the C ABI must be contiguous"
    return t.data_ptr()
