"""Exception types raised at the drop-in boundary.

Same class names and hierarchy as the reference (``gym_anm/errors.py:4-46`` and
``gym_anm/simulator/components/errors.py:1-62``) so that ``except`` clauses written against
gym-anm keep working, plus :class:`HipExtensionError` for the one failure mode the reference
cannot have: the native MI355X library being absent or unusable (there is no CPU fallback).
"""


class ANMEnvConfigurationError(Exception):
    """Problem while constructing an environment."""


class ArgsError(ANMEnvConfigurationError):
    """An environment argument is invalid."""


class ObsSpaceError(ANMEnvConfigurationError):
    """The observation space is not properly specified."""


class ObsNotSupportedError(ObsSpaceError):
    def __init__(self, wanted, allowed):
        super().__init__("Observation type unsupported. Desired {} but we only support {}.".format(wanted, allowed))


class UnitsNotSupportedError(ObsSpaceError):
    def __init__(self, wanted, allowed, key):
        super().__init__(
            "Observation unit unsupported. Desired: {} but we only support {} for observation {}.".format(
                wanted, allowed, key
            )
        )


class EnvInitializationError(ANMEnvConfigurationError):
    """reset() could not find a non-terminal initial state."""


class EnvNextVarsError(ANMEnvConfigurationError):
    """next_vars() returned something of the wrong shape."""


class InputNetworkFileError(Exception):
    """Base class for problems with the network input dictionary."""

    def __init__(self, message=""):
        super().__init__(message)


class BaseMVAError(InputNetworkFileError):
    def __init__(self):
        super().__init__("The network baseMVA should be > 0.")


class BranchSpecError(InputNetworkFileError):
    pass


class BusSpecError(InputNetworkFileError):
    pass


class DeviceSpecError(InputNetworkFileError):
    pass


class GenSpecError(DeviceSpecError):
    pass


class LoadSpecError(DeviceSpecError):
    pass


class StorageSpecError(DeviceSpecError):
    pass


class PFEError(Exception):
    """No solution to the network equations was found."""


class UnsupportedNetworkError(InputNetworkFileError):
    """The network is valid for the reference's parser but not for its power-flow solver.

    ``solve_load_flow.py:32-39,116,157-160,171`` hard-wires the slack bus to sorted position 0 and
    needs bus IDs ``0..N-1``; the reference silently computes garbage (or crashes inside SciPy)
    otherwise.  This build refuses such networks at construction time instead.
    """


class HipExtensionError(RuntimeError):
    """The native gfx950 library is missing, failed to build/load, or no MI355X is visible."""
