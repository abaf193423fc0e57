"""MI355X-native seamless clone (Poisson image editing, NORMAL_CLONE).

Product code only: the HIP shared library (csrc/ -> libseamlessclone_hip.so), its ctypes
binding (capi), the reference-shaped Python class (seamless_clone), OpenCV-yml/BMP I/O (ymlio)
and the vs.py-equivalent checker (compare).  Nothing here imports torch or the CPU oracle.
"""
from . import capi, compare, ymlio  # noqa: F401
from .seamless_clone import SeamlessClone, seamlessClone  # noqa: F401

__all__ = ["capi", "compare", "ymlio", "SeamlessClone", "seamlessClone"]
