"""tbsim.evaluation.policy_composers -> trace_amd.policy (reference policy_composers.py:27-57)."""
from trace_amd.policy import Diffuser, DiffuserTrafficModel, PolicyWrapper  # noqa: F401
