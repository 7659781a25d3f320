"""torchlsq -- MI355X-native LSQ / LSQ+ fake quantization (drop-in for DeadAt0m/LSQFakeQuantize-PyTorch).

    import torchlsq                         # registers torch.ops.torchlsq.* on top of liblsq_hip.so
    from torchlsq.functional import lsq
    from torchlsq.quantized import LSQFakeQuantizer
"""
from .extension import _HAS_OPS  # noqa: F401  (importing it registers the ops)

__version__ = "2.1+mi355x.1"

from torchlsq.quantized import *  # noqa: F401,F403,E402  (reference torchlsq/__init__.py:17)
