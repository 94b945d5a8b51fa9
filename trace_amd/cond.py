"""Once-per-call conditioning of the sampler: ego / neighbour
history MLPs and the condition-fusion MLP, run with PyTorch-ROCm ops (amortised over the 100
diffusion steps; not a kernel target).  Reference: tbsim/models/trace.py:243-339,
tbsim/models/trace_helpers.py:66-174, tbsim/models/base_models.py:14-104.

The ResNet-18 map encoder cannot be instantiated offline (torchvision is absent), so the
feature grid ``map_grid_feat`` (B, 32, 56, 56) is an input of the batch dict; a caller with a
map encoder passes it via ``map_encoder=`` to ``TraceSampler``.
"""
import torch
import torch.nn.functional as F


class Conditioner:
    def __init__(self, cfg, state_dict, device):
        self.cfg = cfg
        self.device = device
        keep = ("agent_hist_encoder.", "neighbor_hist_encoder.", "process_cond_mlp.")
        self.w = {k: v.detach().to(device=device, dtype=torch.float32).contiguous()
                  for k, v in state_dict.items() if k.startswith(keep)}
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=device)
        self.a_add, self.a_div = t(cfg.agent_hist_norm_add), t(cfg.agent_hist_norm_div)
        self.n_add, self.n_div = t(cfg.neighbor_hist_norm_add), t(cfg.neighbor_hist_norm_div)

    def _mlp(self, prefix, x):
        i = 0
        w = self.w
        while ("%s.%d.weight" % (prefix, i + 1)) in w and w["%s.%d.weight" % (prefix, i + 1)].dim() == 1:
            x = F.linear(x, w["%s.%d.weight" % (prefix, i)], w["%s.%d.bias" % (prefix, i)])
            x = F.layer_norm(x, x.shape[-1:], w["%s.%d.weight" % (prefix, i + 1)],
                             w["%s.%d.bias" % (prefix, i + 1)])
            x = F.relu(x)
            i += 3
        return F.linear(x, w["%s.%d.weight" % (prefix, i)], w["%s.%d.bias" % (prefix, i)])

    @staticmethod
    def _features(pos, yaw, speed, extent, avail, add, div):
        n, th, _ = pos.shape
        heading = torch.cat([torch.cos(yaw), torch.sin(yaw)], dim=-1)
        lw = extent[:, :2].unsqueeze(1).expand(n, th, 2)
        pos = (pos + add[:2]) / div[:2]
        speed = (speed.unsqueeze(-1) + add[2]) / div[2]
        lw = (lw + add[3:]) / div[3:]
        f = torch.cat([pos, heading, speed, lw, avail.unsqueeze(-1).to(pos.dtype)], dim=-1)
        return torch.where(avail.unsqueeze(-1), f, torch.zeros_like(f)).reshape(n, -1)

    def _ego(self, b, avail):
        f = self._features(b["history_positions"], b["history_yaws"], b["history_speeds"], b["extent"],
                           avail, self.a_add, self.a_div)
        return self._mlp("agent_hist_encoder.traj_mlp._model", f)

    def _neighbors(self, b, avail):
        pos = b["all_other_agents_history_positions"]
        B, Nn, Th, _ = pos.shape
        f = self._features(pos.reshape(B * Nn, Th, 2),
                           b["all_other_agents_history_yaws"].reshape(B * Nn, Th, 1),
                           b["all_other_agents_history_speeds"].reshape(B * Nn, Th),
                           b["all_other_agents_extents"].reshape(B * Nn, 3),
                           avail.reshape(B * Nn, Th), self.n_add, self.n_div)
        enc = self._mlp("neighbor_hist_encoder.agt_hist_encoder.traj_mlp._model", f).reshape(B, Nn, -1)
        padded = avail.sum(dim=-1) == 0
        enc = enc.masked_fill(padded.unsqueeze(-1), float("-inf"))
        pooled = enc.amax(dim=1)
        alone = padded.sum(dim=-1) == Nn
        return pooled.masked_fill(alone.unsqueeze(-1), 0.0)

    @torch.no_grad()
    def __call__(self, batch, use_cfg):
        """Returns cond_feat (B, cond_dim) and, with classifier-free guidance, the
        unconditional features (all availabilities False, trace.py:287-295,309-317)."""
        ha = batch["history_availabilities"].bool()
        na = batch["all_other_agents_history_availabilities"].bool()
        cond = self._mlp("process_cond_mlp._model",
                         torch.cat([self._ego(batch, ha), self._neighbors(batch, na)], dim=-1))
        non_cond = None
        if use_cfg:
            non_cond = self._mlp("process_cond_mlp._model",
                                 torch.cat([self._ego(batch, torch.zeros_like(ha)),
                                            self._neighbors(batch, torch.zeros_like(na))], dim=-1))
        return cond, non_cond
