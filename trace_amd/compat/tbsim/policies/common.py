"""tbsim.policies.common -> trace_amd.policy (reference common.py:8-64,107-180)."""
from trace_amd.policy import Action, RolloutAction  # noqa: F401
