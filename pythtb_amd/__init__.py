"""pythtb_amd: MI355X-native k-mesh solve + Berry path with PythTB's tb_model / wf_array API.

    from pythtb_amd import *          # instead of: from pythtb import *

Host code is pure Python + NumPy; all k-space work (H(k) assembly, Hermitian
eigen-solves, periodic images, link overlaps, plaquette fluxes, Wilson loops) runs
in hand-written HIP kernels for gfx950 reached through the C ABI of include/tbk.h.
There is no CPU fallback.
"""
from .model import tb_model
from .wfarray import wf_array
from .w90 import w90
from . import shard
from .topology import z2_from_wilson_centres

__version__ = "0.1.0"
__all__ = ["tb_model", "wf_array", "w90", "shard", "z2_from_wilson_centres", "__version__"]
