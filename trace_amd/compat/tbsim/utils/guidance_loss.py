"""tbsim.utils.guidance_loss -> trace_amd.guidance (reference guidance_loss.py:70-105,866-928)."""
from trace_amd.guidance import (DiffuserGuidance, GuidanceConfig, GUIDANCE_NAMES,  # noqa: F401
                                choose_action_from_guidance, verify_guidance_config_list)
from trace_amd.policy import choose_action_from_gt  # noqa: F401
