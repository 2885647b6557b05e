"""moka_hip -- host-side mirror of the MOKA.jl forward-model interface over libmoka_hip.so.

Python stands in for the Julia shim (mpas-ocean.jl_amd/julia/MokaHIP.jl), which cannot be executed
in this pipeline (no Julia).  Names, argument order and error behaviour follow the reference
(src/MOKA.jl:3-16 exports); `!` is dropped from function names.
"""
from . import lib, meshgen  # noqa: F401
from .api import *  # noqa: F401,F403
