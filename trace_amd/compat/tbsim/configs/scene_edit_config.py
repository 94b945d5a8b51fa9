"""tbsim.configs.scene_edit_config -> trace_amd.eval_config (reference scene_edit_config.py:11-138)."""
from trace_amd.eval_config import SceneEditingConfig  # noqa: F401
