"""TRACE guided sampler on MI355X: host driver of the HIP kernels in libtrace_hip.so.

``TraceSampler`` mirrors the reference's ``DiffuserModel`` sampling interface
(/root/reference/tbsim/models/trace.py:214-229 guidance setters, :450-500 ``forward``,
:635-766 ``p_sample`` / ``p_sample_loop``): same argument names, same output dict.  All
per-step work runs in hand-written kernels through the C ABI (include/trace_hip.h); PyTorch
is used for device memory, streams and the once-per-call conditioning MLPs.

There is no CPU fallback: constructing a sampler without the built library or without a GPU
raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _cabi
from ._cabi import DynParams, StepCoefs, UnetConfig, ptr
from .cond import Conditioner
from .config import ModelConfig
from .guidance import DiffuserGuidance, verify_guidance_config_list
from .schedule import Schedule
from .unet_params import flatten_unet_params

C_PAD = 64   # denoiser input channels (38) padded to two 32-channel MFMA k-slices (pad stays zero)


class TraceSampler:
    def __init__(self, cfg: ModelConfig, state_dict, device="cuda:0", map_encoder=None):
        self.lib = _cabi.lib()          # raises if the HIP library is not built
        if not torch.cuda.is_available():
            raise _cabi.TraceHipError("TraceSampler needs a GPU (torch.cuda.is_available() is False); "
                                      "there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.map_encoder = map_encoder
        self.schedule = Schedule(cfg.n_diffusion_steps)
        self.current_guidance = None
        dims = cfg.unet_dims
        assert len(dims) == 4, "the HIP denoiser is built for 3 resolution levels"
        uc = UnetConfig(horizon=cfg.horizon, in_channels=cfg.unet_in_dim, n_levels=3,
                        dims=(C.c_int * 4)(dims[1], dims[2], dims[3], 0), cond_dim=cfg.cond_feat_dim,
                        time_dim=cfg.time_dim, n_groups=cfg.n_groups, n_steps=cfg.n_diffusion_steps)
        flat = flatten_unet_params(cfg, state_dict)
        n_expected = self.lib.trace_unet_param_count(C.byref(uc))
        if flat.numel() != n_expected:
            raise _cabi.TraceHipError("denoiser parameter count %d != expected %d"
                                      % (flat.numel(), n_expected))
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _cabi.check(self.lib.trace_unet_create(C.byref(uc), flat.data_ptr(), flat.numel(),
                                                   C.byref(handle)), "trace_unet_create")
        self._unet = handle
        self.cond_width = self.lib.trace_unet_cond_width(self._unet)
        self.conditioner = Conditioner(cfg, state_dict, self.device)
        p = DynParams()
        for j in range(6):
            p.norm_add[j] = np.float32(cfg.norm_add[j])
            p.norm_div[j] = np.float32(cfg.norm_div[j])
        p.dt = cfg.step_time
        p.acc_lo, p.acc_hi = cfg.acce_bound
        p.v_lo, p.v_hi = cfg.vbound
        p.max_steer, p.max_yawvel = cfg.max_steer, cfg.max_yawvel
        self._dyn = p
        self._bufs = {}
        self._graphs = {}
        add = torch.tensor(cfg.norm_add, dtype=torch.float32, device=self.device)
        div = torch.tensor(cfg.norm_div, dtype=torch.float32, device=self.device)
        self._add, self._div = add, div

    def __del__(self):
        try:
            if getattr(self, "_unet", None):
                self.lib.trace_unet_destroy(self._unet)
                self._unet = None
        except Exception:
            pass

    # ---- guidance setters (trace.py:214-229) ------------------------------------------- #
    def set_guidance(self, guidance_config_list, example_batch=None):
        if guidance_config_list is not None:
            if len(guidance_config_list) > 0 and verify_guidance_config_list(guidance_config_list):
                self.current_guidance = DiffuserGuidance(guidance_config_list, example_batch)

    def update_guidance(self, **kwargs):
        if self.current_guidance is not None:
            self.current_guidance.update(**kwargs)

    def clear_guidance(self):
        self.current_guidance = None

    # ---- helpers ------------------------------------------------------------------------ #
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _coefs(self, t):
        c = self.schedule.coefs(t)
        sc = StepCoefs()
        for k, v in c.items():
            setattr(sc, k, v)
        return sc

    def descale_traj(self, x):
        return x * self._div - self._add

    def scale_traj(self, x):
        return (x + self._add) / self._div

    def _f32(self, t):
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def get_aux_info(self, data_batch, include_class_free_cond=False):
        """trace.py:243-339 (conditioning; map encoder output taken from the batch)."""
        b = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data_batch.items()}
        cond, non_cond = self.conditioner(b, include_class_free_cond)
        B = cond.shape[0]
        curr = torch.zeros(B, 4, device=self.device)
        curr[:, 2] = b["curr_speed"]
        if "map_grid_feat" in b:
            grid = b["map_grid_feat"]
            grid_nc = b.get("map_grid_feat_non_cond")
        elif self.map_encoder is not None:
            _, grid = self.map_encoder(b["image"])
            grid_nc = None
            if include_class_free_cond:
                _, grid_nc = self.map_encoder(torch.ones_like(b["image"]) * self.cfg.conditioning_drop_fill)
        else:
            raise KeyError("data_batch has no 'map_grid_feat' and no map_encoder was given "
                           "(the ResNet-18 map encoder is outside this build, see DESIGN.md)")
        aux = {"cond_feat": cond, "curr_states": curr.contiguous(),
               "map_grid_feat": grid, "raster_from_agent": self._f32(b["raster_from_agent"])}
        if include_class_free_cond:
            assert grid_nc is not None, "classifier-free guidance needs 'map_grid_feat_non_cond'"
            aux["non_cond_feat"] = non_cond
            aux["map_grid_feat_non_cond"] = grid_nc
        return aux

    # ---- the sampling loop ----------------------------------------------------------------- #
    @torch.no_grad()
    def forward(self, data_batch, num_samp=1, return_diffusion=False, return_guidance_losses=False,
                class_free_guide_w=0.0, apply_guidance=True, guide_clean=False,
                noise_init=None, noise_steps=None, generator=None, n_steps=None, x_init=None,
                steps=None, use_graph=False):
        """DiffuserModel.forward (trace.py:450-500).  Extra keyword arguments:
        noise_init (B,N,T,6) / noise_steps (S,B*N,T,2): injected noise in the reference's draw
        order (trace.py:721,655), else drawn on the device; n_steps: run only the first n reverse
        steps (profiling); x_init (B*N,T,6) + steps [t,...]: start from a given scaled state and
        run the listed diffusion steps (teacher-forced parity tests); use_graph: capture the
        step loop into a hipGraph on first use and replay it afterwards (launch-bound batches)."""
        with torch.cuda.device(self.device):
            return self._forward(data_batch, num_samp, return_diffusion, return_guidance_losses,
                                 float(class_free_guide_w), apply_guidance, guide_clean, noise_init,
                                 noise_steps, generator, n_steps, x_init, steps, use_graph)

    __call__ = forward

    def _forward(self, data_batch, N, return_diffusion, return_guidance_losses, w, apply_guidance,
                 guide_clean, noise_init, noise_steps, generator, n_steps, x_init=None, steps=None,
                 use_graph=False):
        cfg, L, dev = self.cfg, self.lib, self.device
        st = self._stream()
        use_cfg = w != 0.0
        aux = self.get_aux_info(data_batch, use_cfg)
        B = aux["cond_feat"].shape[0]
        T = cfg.horizon
        M = B * N
        R = 2 * M if use_cfg else M
        dyn = C.byref(self._dyn)
        Cg = cfg.map_grid_feature_dim
        Hg, Wg = aux["map_grid_feat"].shape[-2:]
        # Everything the step loop reads lives in buffers owned by the sampler (one per name and
        # shape), so a captured hipGraph of the loop can be replayed on the next call's inputs.
        curr = self._buf("curr", (B, 4))
        curr.copy_(aux["curr_states"])
        rfa = self._buf("rfa", (B, 3, 3))
        rfa.copy_(aux["raster_from_agent"])
        grid = self._buf("grid", (B, Hg, Wg, Cg))
        grid.copy_(self._f32(aux["map_grid_feat"]).permute(0, 2, 3, 1))
        grid_nc = None
        if use_cfg:
            grid_nc = self._buf("grid_nc", (B, Hg, Wg, Cg))
            grid_nc.copy_(self._f32(aux["map_grid_feat_non_cond"]).permute(0, 2, 3, 1))
        # per-block additive cond vectors (temporal.py:22-26,44)
        cond_rows = torch.cat([aux["cond_feat"], aux["non_cond_feat"]], 0) if use_cfg else aux["cond_feat"]
        cond_rows = cond_rows.contiguous()
        table = self._buf("cond_table", (cond_rows.shape[0], self.cond_width))
        _cabi.check(L.trace_unet_cond_table(self._unet, ptr(cond_rows), cond_rows.shape[0], ptr(table), st),
                    "trace_unet_cond_table")

        guidance = self.current_guidance
        plan = guidance.build_plan(data_batch, N, T, dev, generator) if guidance is not None else None
        if plan is not None and getattr(self, "debug_map_argmin", False):
            plan.debug_map_argmin = True         # tests: out["guidance_plan"].map_argmin (guidance.py)
        guided = plan is not None and apply_guidance
        save = 1 if (guided and guide_clean) else 0
        ws_bytes = L.trace_unet_workspace_bytes(self._unet, R, save)
        ws = self._buf("unet_ws", (ws_bytes,), torch.uint8)
        x_in = self._buf("x_in", (R, C_PAD, T), zero=True)
        pred = self._buf("pred", (R, T, 2))
        x0_buf = self._buf("x0", (M, T, 2))
        xa = self._buf("xa", (M, T, 6))
        xb = self._buf("xb", (M, T, 6))
        a_out = self._buf("a_out", (M, T, 2))
        traj = self._buf("traj", (M, T, 6)) if plan is not None else None
        gtraj = self._buf("gtraj", (M, T, 6)) if guided else None
        g_act = self._buf("g_act", (M, T, 2)) if guided else None
        mean = self._buf("mean", (M, T, 2)) if guided and not guide_clean else None
        gpred = self._buf("gpred", (R, T, 2)) if save else None
        gx = self._buf("gx", (R, T, 2)) if save else None
        img_w, img_h = float(cfg.image_shape[2]), float(cfg.image_shape[1])

        # x_T: only the action channels of the initial noise are used (trace.py:721,731)
        if x_init is not None:
            xa.copy_(self._f32(x_init).reshape(M, T, 6))
        else:
            if noise_init is None:
                noise_init = torch.randn(B, N, T, 6, device=dev, generator=generator)
            a0 = self._f32(noise_init).reshape(M, T, 6)[..., 4:6].contiguous()
            _cabi.check(L.trace_dyn_rollout(ptr(a0), ptr(curr), ptr(xa), M, N, T, 1, 1, dyn, st),
                        "trace_dyn_rollout")
        if steps is None:
            steps = list(reversed(range(cfg.n_diffusion_steps)))
            if n_steps is not None:
                steps = steps[:n_steps]
        steps = [int(t) for t in steps]
        S = len(steps)
        diffusion = None
        if return_diffusion:
            # the chain x_T .. x_0 is kept: step k reads diffusion[k] and writes diffusion[k + 1] in place of the
            # xa / xb ping-pong (no copy per step)
            diffusion = self._buf("diffusion", (S + 1, M, T, 6))
            diffusion[0].copy_(xa)
        # all step noise up front (the reference draws one randn_like per step, trace.py:655; an
        # injected tape keeps its order): the loop itself is then free of RNG state
        noise = self._buf("noise", (S, M, T, 2))
        if noise_steps is not None:
            noise.copy_(self._f32(noise_steps)[:S].reshape(S, M, T, 2))
        else:
            noise.normal_(generator=generator)

        def run_loop(plan_):
            st_ = self._stream()                 # the capturing stream when under graph capture
            x, x_next = xa, xb
            losses = None
            # without classifier-free guidance x0 IS the denoiser's prediction (trace_cfg_combine would only copy it)
            x0 = pred if (not use_cfg and w == 0.0) else x0_buf
            for k, t in enumerate(steps):
                sc = self._coefs(t)
                csc = C.byref(sc)
                if diffusion is not None:
                    x, x_next = diffusion[k], diffusion[k + 1]
                # denoiser input: trajectory + bilinear map features, cond and uncond rows stacked
                if use_cfg:
                    _cabi.check(L.trace_map_query_cfg(ptr(x), ptr(grid), ptr(grid_nc), ptr(rfa), ptr(x_in), M, N,
                                                      T, Cg, Hg, Wg, C_PAD, img_w, img_h, dyn, st_),
                                "trace_map_query_cfg")
                else:
                    _cabi.check(L.trace_map_query(ptr(x), ptr(grid), ptr(rfa), ptr(x_in), M, N, T, Cg, Hg, Wg,
                                                  C_PAD, img_w, img_h, dyn, st_), "trace_map_query")
                _cabi.check(L.trace_unet_forward(self._unet, ptr(x_in), ptr(table), N, t, R, ptr(pred),
                                                 ptr(ws), ws_bytes, save, st_), "trace_unet_forward")
                if x0 is not pred:
                    _cabi.check(L.trace_cfg_combine(ptr(pred), pred[M:].data_ptr() if use_cfg else None,
                                                    ptr(x), ptr(x0), M, T, w, csc, st_), "trace_cfg_combine")
                grad = None
                if guided:
                    if guide_clean:
                        xg = x0
                    else:
                        _cabi.check(L.trace_posterior_mean(ptr(x0), ptr(x), ptr(mean), M, T, csc, st_),
                                    "trace_posterior_mean")
                        xg = mean
                    _cabi.check(L.trace_dyn_rollout(ptr(xg), ptr(curr), ptr(traj), M, N, T, 1, 0, dyn, st_),
                                "trace_dyn_rollout")
                    gtraj.zero_()
                    plan_.evaluate(traj, gtraj, st_)
                    _cabi.check(L.trace_dyn_adjoint(ptr(xg), ptr(curr), ptr(gtraj), ptr(g_act), M, N, T,
                                                    dyn, st_), "trace_dyn_adjoint")
                    if guide_clean:
                        _cabi.check(L.trace_cfg_combine_adjoint(
                            ptr(g_act), ptr(gpred), gpred[M:].data_ptr() if use_cfg else None, M, T, w,
                            csc, st_), "trace_cfg_combine_adjoint")
                        _cabi.check(L.trace_unet_backward_input(self._unet, ptr(gpred), R, ptr(gx), ptr(ws),
                                                                ws_bytes, st_), "trace_unet_backward_input")
                        grad = gx
                    else:
                        grad = g_act
                clean_path = 1 if (plan_ is not None and guide_clean) else 0
                if grad is not None and grad is gx and use_cfg:      # both halves of the input gradient, summed inside the kernel
                    _cabi.check(L.trace_posterior_step_cfg(ptr(x0), ptr(gx), gx[M:].data_ptr(), ptr(x),
                                                           ptr(noise[k]), ptr(curr), ptr(x_next), ptr(a_out), M, N,
                                                           T, clean_path, csc, dyn, st_),
                                "trace_posterior_step_cfg")
                else:
                    _cabi.check(L.trace_posterior_step(ptr(x0), ptr(grad) if grad is not None else None,
                                                       ptr(x), ptr(noise[k]), ptr(curr), ptr(x_next),
                                                       ptr(a_out), M, N, T, clean_path, csc, dyn, st_),
                                "trace_posterior_step")
                if plan_ is not None and k == S - 1:
                    # guidance losses of the final sample, for filtering (trace.py:697-700)
                    _cabi.check(L.trace_dyn_rollout(ptr(a_out), ptr(curr), ptr(traj), M, N, T, 1, 0, dyn,
                                                    st_), "trace_dyn_rollout")
                    losses = plan_.evaluate(traj, None, st_)
                x, x_next = x_next, x
            return x, losses

        if use_graph:
            x, layer_losses, plan = self._run_graphed(run_loop, plan, (
                B, N, use_cfg, w, guided, guide_clean, tuple(steps), return_diffusion,
                None if plan is None else plan.signature()))
        else:
            x, layer_losses = run_loop(plan)

        pred_traj = x.reshape(B, N, T, 6).clone()
        tr = self.descale_traj(pred_traj)
        out = {
            "trajectories": tr[..., [0, 1, 3]],
            "predictions": {"positions": tr[..., :2], "yaws": tr[..., 3:4]},
            "curr_states": curr.clone(),
            "pred_traj_scaled": pred_traj,
        }
        if return_diffusion:
            d = diffusion.reshape(S + 1, B, N, T, 6).permute(1, 2, 3, 0, 4)
            out["predictions"]["diffusion_steps"] = self.descale_traj(d)
        if plan is not None:
            if use_graph:
                layer_losses = [l.clone() for l in layer_losses]
            out["guidance_plan"] = plan
            out["guidance_layer_losses"] = layer_losses
            if return_guidance_losses:
                out["predictions"]["guide_losses"] = plan.guide_losses(layer_losses)
        return out

    def _run_graphed(self, run_loop, plan, key):
        """hipGraph replay of the whole step loop (≈45-95 launches per step x 100 steps become one
        graph launch): small batches are launch-bound (cfg1: 1.8 ms of host time per 0.4 ms of GPU
        work per step).  The first call with a given (shape, schedule, guidance structure) key runs
        eagerly and captures; later calls copy their per-call tables into the captured plan and
        replay."""
        entry = self._graphs.get(key)
        if entry is None:
            x, losses = run_loop(plan)                       # eager: also sets kernel attributes
            torch.cuda.synchronize(self.device)
            result = (x.clone(), None if losses is None else [l.clone() for l in losses])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gx_, glosses = run_loop(plan)
            self._graphs[key] = (g, gx_, glosses, plan)
            # the capture did not execute anything; hand back the eager result
            return result[0], result[1], plan
        g, gx_, glosses, gplan = entry
        if plan is not None:
            gplan.load_tables_from(plan)
        g.replay()
        return gx_, glosses, gplan

    # ---- single pieces, exposed for tests / teacher-forced parity -------------------------- #
    @torch.no_grad()
    def denoise(self, x, t, aux, num_samp, w=0.0, save=False):
        """One denoiser evaluation + CFG combine for a given scaled x (M,T,6) and step t.
        Returns (x0, pred) -- p_mean_variance up to x_recon (trace.py:561-598)."""
        with torch.cuda.device(self.device):
            cfg, L, st = self.cfg, self.lib, self._stream()
            x = self._f32(x)
            M, T = x.shape[0], cfg.horizon
            use_cfg = w != 0.0
            R = 2 * M if use_cfg else M
            cond_rows = torch.cat([aux["cond_feat"], aux["non_cond_feat"]], 0) if use_cfg else aux["cond_feat"]
            cond_rows = cond_rows.contiguous()
            table = torch.empty(cond_rows.shape[0], self.cond_width, device=self.device)
            _cabi.check(L.trace_unet_cond_table(self._unet, ptr(cond_rows), cond_rows.shape[0], ptr(table),
                                                st), "trace_unet_cond_table")
            x_in = torch.zeros(R, C_PAD, T, device=self.device)
            Hg, Wg = aux["map_grid_feat"].shape[-2:]
            dyn = C.byref(self._dyn)
            grids = [aux["map_grid_feat"]] + ([aux["map_grid_feat_non_cond"]] if use_cfg else [])
            for i, g in enumerate(grids):
                gl = self._f32(g).permute(0, 2, 3, 1).contiguous()
                _cabi.check(L.trace_map_query(ptr(x), ptr(gl), ptr(aux["raster_from_agent"]),
                                              x_in[i * M:].data_ptr(), M, num_samp, T,
                                              cfg.map_grid_feature_dim, Hg, Wg, C_PAD,
                                              float(cfg.image_shape[2]), float(cfg.image_shape[1]), dyn,
                                              st), "trace_map_query")
            ws_bytes = L.trace_unet_workspace_bytes(self._unet, R, 1 if save else 0)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            pred = torch.empty(R, T, 2, device=self.device)
            _cabi.check(L.trace_unet_forward(self._unet, ptr(x_in), ptr(table), num_samp, int(t), R,
                                             ptr(pred), ptr(ws), ws_bytes, 1 if save else 0, st),
                        "trace_unet_forward")
            x0 = torch.empty(M, T, 2, device=self.device)
            sc = self._coefs(int(t))
            _cabi.check(L.trace_cfg_combine(ptr(pred), pred[M:].data_ptr() if use_cfg else None, ptr(x),
                                            ptr(x0), M, T, w, C.byref(sc), st), "trace_cfg_combine")
            return x0, pred, x_in, (ws, ws_bytes)
