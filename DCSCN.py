"""Drop-in ``DCSCN`` module: ``DCSCN.SuperResolution(flags, model_name)`` as sr.py / evaluate.py use it
(reference DCSCN.py:28), backed by the hand-written gfx950 kernels.  See
dcscn-super-resolution_amd/model.py."""

import dcscn_amd  # noqa: F401
from dcscn_amd.model import BICUBIC_METHOD_STRING, SuperResolution          # noqa: F401
