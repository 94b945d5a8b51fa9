"""tbsim.policies.wrappers -> trace_amd.policy (reference wrappers.py:6-65)."""
from trace_amd.policy import PolicyWrapper, RolloutWrapper  # noqa: F401
