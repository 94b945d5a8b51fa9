"""Import harness for the reference sampler -- TEST INFRASTRUCTURE ONLY.

Used only in the build container (where /root/reference is mounted) to (i) pin the
oracle restatement in ``oracle/trace_oracle.py`` and (ii) emit golden vectors with
``oracle/gen_golden.py``.  Nothing here travels to the GPU box in a useful form:
every entry point raises when /root/reference is absent, and no product code
imports this module.

The recipe follows SURVEY.md section 8(c):
  * sys.modules stubs for torchvision / shapely / trajdata (absent here; the hot
    path only needs them to be importable),
  * ``torch.Tensor.get_device`` shim (the reference's CPU path raises
    "Device index must not be negative" at tbsim/models/trace.py:424-426,441-443),
  * a stand-in for the ResNet-18 ``MapEncoder`` (torchvision is absent) that returns
    preset feature grids for the conditional / filled ("non-cond") image,
  * ``torch.randn`` / ``torch.randn_like`` replaced by a seeded tape while sampling.
"""
import contextlib
import enum
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("TRACE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "tbsim"))


def _install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvm = types.ModuleType("torchvision.models")
        tvr = types.ModuleType("torchvision.models.resnet")
        tvf = types.ModuleType("torchvision.models.feature_extraction")

        def _absent(*a, **k):
            raise RuntimeError("torchvision is not available in this container")

        tvr.resnet18 = _absent
        tvr.resnet50 = _absent
        tvf.create_feature_extractor = _absent
        tv.models = tvm
        tvm.resnet = tvr
        tvm.feature_extraction = tvf
        sys.modules.update({
            "torchvision": tv, "torchvision.models": tvm,
            "torchvision.models.resnet": tvr,
            "torchvision.models.feature_extraction": tvf,
        })
    if "shapely" not in sys.modules:
        sh = types.ModuleType("shapely")
        shg = types.ModuleType("shapely.geometry")
        shg.LineString = object
        shg.Polygon = object
        sh.geometry = shg
        sys.modules.update({"shapely": sh, "shapely.geometry": shg})
    if "trajdata" not in sys.modules:
        td = types.ModuleType("trajdata")

        class AgentType(enum.IntEnum):
            UNKNOWN = 0
            VEHICLE = 1
            PEDESTRIAN = 2
            BICYCLE = 3
            MOTORCYCLE = 4

        td.AgentType = AgentType
        sys.modules["trajdata"] = td


_LOADED = {}


def load_reference():
    """Returns a namespace with the reference modules on the hot path."""
    if _LOADED:
        return _LOADED["ns"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # CPU shim, SURVEY.md 8(c) item (2)
    torch.Tensor.get_device = lambda self: self.device
    import tbsim.models.trace as ref_trace
    import tbsim.models.trace_helpers as ref_helpers
    import tbsim.models.temporal as ref_temporal
    import tbsim.utils.guidance_loss as ref_guidance
    import tbsim.utils.geometry_utils as ref_geom

    class PresetMapEncoder(torch.nn.Module):
        """Stand-in for tbsim.models.trace_helpers.MapEncoder (needs torchvision).
        Returns ``grid_cond`` for a regular image and ``grid_noncond`` for the
        constant-filled image the reference builds for classifier-free guidance
        (tbsim/models/trace.py:271-273)."""

        def __init__(self, *a, **k):
            super().__init__()
            self.grid_cond = None
            self.grid_noncond = None
            self.fill_value = None

        def forward(self, image):
            if self.fill_value is not None and bool((image == self.fill_value).all()):
                return None, self.grid_noncond
            return None, self.grid_cond

    ref_trace.MapEncoder = PresetMapEncoder
    # trace_helpers.MapGridDecoder (and base_models.Up / DoubleConv / ConvBlock under it) is pure torch: the
    # decoder half of the map encoder is available from the reference itself (gen_golden_map_decoder.py)
    ns = types.SimpleNamespace(trace=ref_trace, helpers=ref_helpers, temporal=ref_temporal,
                               guidance=ref_guidance, geom=ref_geom,
                               PresetMapEncoder=PresetMapEncoder,
                               MapGridDecoder=ref_helpers.MapGridDecoder)
    _LOADED["ns"] = ns
    return ns


@contextlib.contextmanager
def noise_tape(tensors):
    """Replace torch.randn / torch.randn_like by successive entries of ``tensors``
    (consumption order: SURVEY.md appendix A, tbsim/models/trace.py:721,655)."""
    it = iter(tensors)
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def fake_randn(*shape, **kw):
        t = next(it)
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == shp, (t.shape, shp)
        return t.clone()

    def fake_randn_like(x, **kw):
        t = next(it)
        assert t.shape == x.shape, (t.shape, x.shape)
        return t.clone()

    torch.randn, torch.randn_like = fake_randn, fake_randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like


def build_reference_model(algo, state_dict, grid_cond, grid_noncond, image_shape=(2, 224, 224)):
    """Instantiate the reference DiffuserModel (tbsim/models/trace.py:33) with the
    orca_mixed hyper-parameters in ``algo`` (a dict shaped like the "algo" block of
    configs/train/diffuser_orca_mlp_hist_grid_map_drop_10.json) and load
    ``state_dict`` (reference key names)."""
    ns = load_reference()
    model = ns.trace.DiffuserModel(
        rasterized_map=True, use_map_feat_global=False, use_map_feat_grid=True,
        map_encoder_model_arch="resnet18", input_image_shape=image_shape,
        map_feature_dim=algo["map_feature_dim"], map_grid_feature_dim=algo["map_grid_feature_dim"],
        hist_num_frames=algo["history_num_frames"] + 1, hist_feature_dim=algo["history_feature_dim"],
        cond_feature_dim=algo["cond_feat_dim"], diffuser_model_arch="TemporalMapUnet",
        horizon=algo["horizon"], observation_dim=4, action_dim=2, output_dim=2,
        n_timesteps=algo["n_diffusion_steps"], loss_type="l2", action_weight=1, loss_discount=1,
        dim_mults=tuple(algo["dim_mults"]), dynamics_type="Unicycle", dynamics_kwargs=algo["dynamics"],
        base_dim=algo["base_dim"], diffuser_input_mode="state_and_action", use_conditioning=True,
        cond_fill_value=algo["conditioning_drop_fill"], diffuser_norm_info=algo["diffuser_norm_info"],
        agent_hist_norm_info=algo["agent_hist_norm_info"],
        neighbor_hist_norm_info=algo["neighbor_hist_norm_info"], dt=algo["step_time"])
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    # schedule buffers / loss weights are derived, not learned (trace.py:177-207)
    missing = [k for k in missing if not k.startswith("map_encoder") and "." in k
               and not k.startswith("loss_fn.")]
    assert not missing, missing
    assert not unexpected, unexpected
    model.map_encoder.grid_cond = grid_cond
    model.map_encoder.grid_noncond = grid_noncond
    model.map_encoder.fill_value = algo["conditioning_drop_fill"]
    model.eval()
    model.requires_grad_(False)
    return model
