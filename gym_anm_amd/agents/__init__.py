"""Batched MPC baseline policies (the reference's ``gym_anm/agents``)."""
from .mpc import MPCAgent, MPCAgentConstant, MPCAgentPerfect  # noqa: F401
