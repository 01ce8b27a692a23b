"""gym_anm_amd -- MI355X-native batched implementation of gym-anm's simulator step.

Importing the package never touches the GPU; the HIP extension is loaded on first use and a
missing / unloadable extension raises (there is no CPU fallback in the product path).
"""

__version__ = "0.1.0"
