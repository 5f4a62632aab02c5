"""``cross_entropy`` / ``CrossEntropyLoss``: the ``nn.CrossEntropyLoss()`` of the train step (reference main.py:94,
:214; mean over the batch) through the C-ABI of ``include/coattn.h`` (``csrc/ce.hip``) on the caller's current
stream: loss and d loss / d logits in one pass.  SURVEY.md section 8f-1.  The ``MLPClassifier`` that produces the
logits (model.py:400-434) is the stock module of ``modules.py``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_last = None          # (ws, B, device) of the last call: check_labels() reads its status word


class _CrossEntropyFn(torch.autograd.Function):
    """Mean cross entropy; the forward pass also produces d loss / d logits."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, logits, labels):
        if not logits.is_cuda:
            raise RuntimeError("cross_entropy (HIP) needs tensors on the GPU")
        if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0] or labels.dtype != torch.int64:
            raise RuntimeError("cross_entropy: logits [B,K] fp32 and labels [B] int64 expected")
        lib = _lib.load()
        B, K = logits.shape
        z = logits.contiguous()
        lab = labels.contiguous()
        dev = z.device
        n = C.c_size_t()
        _lib.check(lib.coattn_ce_workspace_bytes(B, K, _lib.F32, C.byref(n)), "coattn_ce_workspace_bytes")
        ws = torch.empty(n.value // 4, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dz = torch.empty_like(z) if need else None
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with _lib.on_device(dev):
            _lib.check(lib.coattn_ce_forward(_ptr(z), _ptr(lab), _ptr(loss), _ptr(dz), _ptr(ws), B, K, _lib.F32,
                                             stream), "coattn_ce_forward")
        global _last
        _last = (ws, B, dev)
        if need:
            ctx.save_for_backward(dz)
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


def check_labels() -> None:
    """``nn.CrossEntropyLoss`` raises on a label outside [0, K); the HIP kernel stays asynchronous, returns NaN and sets
    a status word instead.  This SYNCHRONISES the current stream and raises IndexError if the last ``cross_entropy``
    call met such a label -- call it where the host synchronises anyway (when the loss is read)."""
    if _last is None:
        return
    ws, B, dev = _last
    lib = _lib.load()
    with _lib.on_device(dev):
        rc = lib.coattn_ce_status(_ptr(ws), B, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc == -2:
        raise IndexError(lib.coattn_last_error().decode())
    _lib.check(rc, "coattn_ce_status")


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``nn.CrossEntropyLoss()(logits, labels)`` (mean reduction, main.py:94/:214) on the HIP path."""
    return _CrossEntropyFn.apply(logits, labels)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for the ``nn.CrossEntropyLoss()`` criterion of the train loop (main.py:94): CUDA fp32 logits take
    the fused HIP kernel, everything else (CPU tensors, the baseline model's CPU runs) the stock functional."""

    def forward(self, logits, labels):
        if logits.is_cuda and logits.dim() == 2 and labels.dtype == torch.int64:
            return cross_entropy(logits, labels)
        return torch.nn.functional.cross_entropy(logits, labels)
