"""gym_anm_amd -- MI355X-native batched implementation of gym-anm's simulator step.

Importing the package never touches the GPU; the HIP extension is loaded on first use and a
missing / unloadable extension raises (there is no CPU fallback in the product path).
"""

__version__ = "0.1.0"

_TOP_LEVEL = {  # the names gym_anm exports at top level (gym_anm/__init__.py:5-6), resolved on first use
    "ANMEnv": ("gym_anm_amd.envs", "ANMEnv"),
    "MPCAgentPerfect": ("gym_anm_amd.agents", "MPCAgentPerfect"),
    "MPCAgentConstant": ("gym_anm_amd.agents", "MPCAgentConstant"),
}


def __getattr__(name):
    if name in _TOP_LEVEL:
        import importlib

        mod, attr = _TOP_LEVEL[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))


def __dir__():
    return sorted(list(globals()) + list(_TOP_LEVEL))


def _register():
    """``gym.make("ANM6Easy-v0")`` like the reference (gym_anm/__init__.py:8-11), when gymnasium is
    installed and the id is still free (gym-anm itself may be installed next to this package)."""
    try:
        from gymnasium.envs.registration import register, registry
    except Exception:
        return
    if "ANM6Easy-v0" not in registry:
        register(id="ANM6Easy-v0", entry_point="gym_anm_amd.envs:ANM6Easy")
    if "ANM6EasyVec-v0" not in registry:
        register(id="ANM6EasyVec-v0", entry_point="gym_anm_amd.envs:ANM6EasyVec")


_register()
