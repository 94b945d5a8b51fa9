"""State-dict layout of the TRACE sampler (reference key names) and a seeded
random-init recipe.

Key names follow the reference's ``DiffuserModel`` modules (SURVEY.md section 5:
``model.downs.{i}.{0,1}.blocks.{0,1}.block.{0,2}``, ``.time_mlp.1``, ``.residual_conv``,
``model.downs.{i}.2.conv``, ``model.ups.{i}.2.conv``, ``model.mid_block{1,2}``,
``model.final_conv.{0.block.{0,2},1}``, ``model.time_mlp.{1,3}``,
``agent_hist_encoder.traj_mlp._model.*``, ``neighbor_hist_encoder.agt_hist_encoder...``,
``process_cond_mlp._model.*``) so that a Lightning checkpoint's ``ema_policy.*`` /
``nets.policy.*`` sub-dict loads unchanged.

No checkpoints ship with the reference (ckpt/README.md), so tests, goldens and the
benchmark use ``random_state_dict(cfg, seed)``: the same recipe feeds the reference
(golden generation), the CPU oracle and the HIP path.
"""
from collections import OrderedDict

import torch

from .config import ModelConfig, unet_block_table


def state_dict_shapes(cfg: ModelConfig) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    hd = cfg.hist_frames * 8
    hf = cfg.history_feature_dim

    def mlp(prefix, dims_in, layer_dims, out_dim):
        d = dims_in
        idx = 0
        for l in layer_dims:
            s["%s.%d.weight" % (prefix, idx)] = (l, d)
            s["%s.%d.bias" % (prefix, idx)] = (l,)
            s["%s.%d.weight" % (prefix, idx + 1)] = (l,)   # LayerNorm
            s["%s.%d.bias" % (prefix, idx + 1)] = (l,)
            idx += 3
            d = l
        s["%s.%d.weight" % (prefix, idx)] = (out_dim, d)
        s["%s.%d.bias" % (prefix, idx)] = (out_dim,)

    mlp("agent_hist_encoder.traj_mlp._model", hd, (hd, hd, hf, hf), hf)
    mlp("neighbor_hist_encoder.agt_hist_encoder.traj_mlp._model", hd, (hd, hd, hf, hf), hf)
    c = 2 * hf
    cd = cfg.cond_feat_dim
    mlp("process_cond_mlp._model", c, (c, c, cd, cd), cd)

    td = cfg.time_dim
    s["model.time_mlp.1.weight"] = (td * 4, td)
    s["model.time_mlp.1.bias"] = (td * 4,)
    s["model.time_mlp.3.weight"] = (td, td * 4)
    s["model.time_mlp.3.bias"] = (td,)

    def conv_block(prefix, ci, co, k=5):
        s[prefix + ".block.0.weight"] = (co, ci, k)
        s[prefix + ".block.0.bias"] = (co,)
        s[prefix + ".block.2.weight"] = (co,)
        s[prefix + ".block.2.bias"] = (co,)

    for prefix, ci, co, _L in unet_block_table(cfg):
        s[prefix + ".time_mlp.1.weight"] = (co, cfg.embed_dim)
        s[prefix + ".time_mlp.1.bias"] = (co,)
        conv_block(prefix + ".blocks.0", ci, co)
        conv_block(prefix + ".blocks.1", co, co)
        if ci != co:
            s[prefix + ".residual_conv.weight"] = (co, ci, 1)
            s[prefix + ".residual_conv.bias"] = (co,)
    dims = cfg.unet_dims
    n_res = len(dims) - 1
    for i in range(n_res - 1):
        co = dims[i + 1]
        s["model.downs.%d.2.conv.weight" % i] = (co, co, 3)
        s["model.downs.%d.2.conv.bias" % i] = (co,)
    for i, ci in enumerate(reversed(dims[1:-1])):
        s["model.ups.%d.2.conv.weight" % i] = (ci, ci, 4)   # ConvTranspose1d: (in, out, k)
        s["model.ups.%d.2.conv.bias" % i] = (ci,)
    fd = dims[1]
    conv_block("model.final_conv.0", fd, fd)
    s["model.final_conv.1.weight"] = (2, fd, 1)
    s["model.final_conv.1.bias"] = (2,)
    return s


def random_state_dict(cfg: ModelConfig, seed: int = 0, gain: float = 1.0):
    """Seeded random weights (torch CPU generator; bitwise reproducible for a fixed
    torch build).  Linear/conv weights ~ U(-b, b), b = gain*sqrt(3/fan_in) (unit-gain
    variance scaling so activations stay O(1) through the 25 conv blocks); norm scales
    1 + 0.1 N(0,1); biases 0.1 N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in state_dict_shapes(cfg).items():
        if len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            if ".2.conv.weight" in k and "ups." in k:
                fan_in = shp[0] * shp[2] // 2      # transposed conv: 2 taps hit each output
            b = gain * (3.0 / fan_in) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        elif k.endswith(".weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(shp, generator=g)
    return sd


def extract_policy_state_dict(ckpt_state_dict, use_ema=True):
    """Sub-dict of a Lightning checkpoint's ``state_dict`` for the sampler
    (reference tbsim/algos/algos.py:54,96: ``nets.policy.*`` / ``ema_policy.*``)."""
    prefix = "ema_policy." if use_ema else "nets.policy."
    out = OrderedDict((k[len(prefix):], v) for k, v in ckpt_state_dict.items()
                      if k.startswith(prefix))
    if not out:
        raise KeyError("no '%s*' keys in checkpoint" % prefix)
    return out
