"""MI355X-native Hierarchical Parallel Co-Attention path (drop-in for the reference's
``--model attention`` hot path, /root/reference/model.py:337-397).

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the compute
is a C-ABI HIP library (``include/coattn.h``, sources in ``csrc/``) loaded through ctypes.
There is no CPU fallback: using the co-attention op without the HIP library or on CPU
tensors raises.
"""
from . import _lib  # noqa: F401
from ._lib import RangeError, check_range  # noqa: F401
from .coattention import ParallelCoAttention, coattention, native_features  # noqa: F401
from .loss import CrossEntropyLoss, cross_entropy  # noqa: F401
from .head import answer_head  # noqa: F401

__all__ = ["ParallelCoAttention", "coattention", "native_features", "answer_head", "cross_entropy", "CrossEntropyLoss", "_lib",
           "check_range", "RangeError"]
