"""tbsim.configs.config -> trace_amd.eval_config (reference config.py:14-190)."""
from trace_amd.eval_config import Dict  # noqa: F401
