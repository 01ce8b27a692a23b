"""Batched (and single-environment) drop-ins for ``gym_anm.envs``."""

from .anm_env import BatchedANMEnv
from .anm6 import ANM6Vec, ANM6EasyVec
from .single import ANM6, ANM6Easy, ANMEnv
from .vector import NumpyVectorEnv
from .mixed import MixedBatchedANMEnv
