"""Canonical flat parameter order expected by ``trace_unet_create`` (include/trace_hip.h).

Order (all fp32, C-contiguous, reference state_dict key names):
  for every residual block in execution order (``config.unet_block_table``):
      <p>.time_mlp.1.weight (C_out, time_dim+cond_dim), <p>.time_mlp.1.bias
      <p>.blocks.0.block.0.{weight (C_out,C_in,5), bias}, <p>.blocks.0.block.2.{weight, bias}
      <p>.blocks.1.block.0.{weight (C_out,C_out,5), bias}, <p>.blocks.1.block.2.{weight, bias}
      <p>.residual_conv.{weight (C_out,C_in,1), bias}            (only if C_in != C_out)
  model.downs.{i}.2.conv.{weight (C,C,3), bias}   i = 0 .. n_levels-2
  model.ups.{i}.2.conv.{weight (C,C,4), bias}     i = 0 .. n_levels-2  (ConvTranspose: in,out,k)
  model.final_conv.0.block.0.{weight, bias}, model.final_conv.0.block.2.{weight, bias}
  model.final_conv.1.{weight (2,C,1), bias}
  model.time_mlp.1.{weight, bias}, model.time_mlp.3.{weight, bias}
"""
import torch

from .config import ModelConfig, unet_block_table


def unet_param_keys(cfg: ModelConfig):
    keys = []
    for p, ci, co, _L in unet_block_table(cfg):
        keys += [p + ".time_mlp.1.weight", p + ".time_mlp.1.bias"]
        for b in (0, 1):
            keys += ["%s.blocks.%d.block.0.weight" % (p, b), "%s.blocks.%d.block.0.bias" % (p, b),
                     "%s.blocks.%d.block.2.weight" % (p, b), "%s.blocks.%d.block.2.bias" % (p, b)]
        if ci != co:
            keys += [p + ".residual_conv.weight", p + ".residual_conv.bias"]
    n_res = len(cfg.unet_dims) - 1
    for i in range(n_res - 1):
        keys += ["model.downs.%d.2.conv.weight" % i, "model.downs.%d.2.conv.bias" % i]
    for i in range(n_res - 1):
        keys += ["model.ups.%d.2.conv.weight" % i, "model.ups.%d.2.conv.bias" % i]
    keys += ["model.final_conv.0.block.0.weight", "model.final_conv.0.block.0.bias",
             "model.final_conv.0.block.2.weight", "model.final_conv.0.block.2.bias",
             "model.final_conv.1.weight", "model.final_conv.1.bias",
             "model.time_mlp.1.weight", "model.time_mlp.1.bias",
             "model.time_mlp.3.weight", "model.time_mlp.3.bias"]
    return keys


def flatten_unet_params(cfg: ModelConfig, state_dict) -> torch.Tensor:
    parts = []
    for k in unet_param_keys(cfg):
        if k not in state_dict:
            raise KeyError("state_dict is missing '%s'" % k)
        parts.append(state_dict[k].detach().to(torch.float32).cpu().contiguous().reshape(-1))
    return torch.cat(parts).contiguous()
