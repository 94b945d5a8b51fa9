"""Diffusion schedule tables, computed the way the reference computes them
(/root/reference/tbsim/models/trace_helpers.py:330-341 cosine betas in float64 numpy ->
float32; /root/reference/tbsim/models/trace.py:170-203 derived tables in float32 torch,
with numpy's float32 sqrt for the two posterior-mean coefficients, which differs from
torch.sqrt by 1 ulp on some entries).  tests/test_host_cpu.py checks them bit-for-bit
against the golden tables.
"""
import numpy as np
import torch


def cosine_betas(n_steps, s=0.008):
    steps = n_steps + 1
    grid = np.linspace(0, steps, steps)
    acp = np.cos(((grid / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    acp = acp / acp[0]
    return torch.tensor(np.clip(1 - (acp[1:] / acp[:-1]), a_min=0, a_max=0.999), dtype=torch.float32)


class Schedule:
    NAMES = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
             "posterior_mean_coef2", "posterior_log_variance_clipped",
             "sqrt_alphas_over_one_minus_alphas_cumprod", "sqrt_recip_one_minus_alphas_cumprod")

    def __init__(self, n_steps):
        self.n_steps = int(n_steps)
        betas = cosine_betas(n_steps)
        alphas = 1.0 - betas
        acp = torch.cumprod(alphas, dim=0)
        prev = torch.cat([torch.ones(1), acp[:-1]])
        var = betas * (1.0 - prev) / (1.0 - acp)
        self.betas = betas
        self.alphas_cumprod = acp
        self.posterior_variance = var
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1.0 / acp)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / acp - 1)
        self.posterior_log_variance_clipped = torch.log(torch.clamp(var, min=1e-20))
        self.posterior_mean_coef1 = betas * torch.from_numpy(np.sqrt(prev.numpy())) / (1.0 - acp)
        self.posterior_mean_coef2 = (1.0 - prev) * torch.from_numpy(np.sqrt(alphas.numpy())) / (1.0 - acp)
        self.sqrt_alphas_over_one_minus_alphas_cumprod = torch.sqrt(acp / (1.0 - acp))
        self.sqrt_recip_one_minus_alphas_cumprod = 1.0 / torch.sqrt(1.0 - acp)

    def coefs(self, t):
        """Scalars of diffusion step t as a dict of Python floats (exact float32 values)."""
        out = {k: float(getattr(self, k)[t]) for k in self.NAMES}
        out["t"] = int(t)
        return out
