"""trace_amd: MI355X-native guided trajectory-diffusion sampler (TRACE hot path).

Importing the package is cheap and GPU-free; the product entry points
(``trace_amd.sampler.TraceSampler``, ``trace_amd.policy``) load libtrace_hip.so and raise
``TraceHipError`` if it is missing -- there is no CPU fallback.
"""
from .config import ModelConfig  # noqa: F401

__all__ = ["ModelConfig"]
