"""tbsim.utils.scene_edit_utils -> trace_amd.scene_edit (reference scene_edit_utils.py:9-392)."""
from trace_amd.scene_edit import (HEURISTIC_FUNC, compute_heuristic_guidance, guided_rollout,  # noqa: F401
                                  merge_guidance_configs)
