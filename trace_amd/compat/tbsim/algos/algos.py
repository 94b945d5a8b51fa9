"""tbsim.algos.algos -> trace_amd.policy (reference algos.py:19-326)."""
from trace_amd.policy import DiffuserTrafficModel  # noqa: F401
