"""csmri_hip: MI355X (gfx950) operator library for the CS-MRI GAN-refinement path.

``lib``  -- ctypes binding of libcsmri_hip.so (C-ABI, include/csmri_hip.h)
``ops``  -- host-side operator layer (autograd glue around the C-ABI calls)
"""
from . import lib          # noqa: F401  (raises ImportError when the .so is missing)
from . import ops          # noqa: F401
from .lib import F32, BF16  # noqa: F401
