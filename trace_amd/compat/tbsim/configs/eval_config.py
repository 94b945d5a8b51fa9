"""tbsim.configs.eval_config -> trace_amd.eval_config (reference eval_config.py:7-87)."""
from trace_amd.eval_config import EvaluationConfig  # noqa: F401
