"""tbsim.envs.env_trajdata -> trace_amd.sim_env: the synthetic, device-resident stand-in for
EnvUnifiedSimulation (reference env_trajdata.py:18-416; datasets are absent offline)."""
from trace_amd.sim_env import SceneBatchEnv as EnvUnifiedSimulation  # noqa: F401
