"""``mlp_classify`` / ``cross_entropy``: the consumer of the co-attention outputs on MI355X --
``MLPClassifier.forward`` (reference model.py:414-434) and the ``nn.CrossEntropyLoss()`` of the train step
(main.py:94, :214) through the C-ABI of ``include/coattn.h`` (``csrc/mlp.hip``) on the caller's current
stream.  SURVEY.md section 8f-1.

The co-attention hands over ``v, q`` as ``[3, B, d]`` buffers; a layer of the head is one GEMM launch that
sums the products of its pieces (no add / concat / tanh passes), and the gradient with respect to ``v`` and
``q`` is one shared ``[3, B, d]`` tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _mlp_bytes(B, d, mlp, K):
    s, f, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _lib.check(_lib.load().coattn_mlp_workspace_bytes(B, d, mlp, K, _lib.F32, C.byref(s), C.byref(f), C.byref(b)),
               "coattn_mlp_workspace_bytes")
    return s.value, f.value, b.value


class _MlpFn(torch.autograd.Function):
    """forward -> coattn_mlp_forward, backward -> coattn_mlp_backward."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # fp32 island under autocast
    def forward(ctx, v, q, W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h):
        if not v.is_cuda:
            raise RuntimeError("mlp_classify (HIP) needs tensors on the GPU")
        if v.dtype != torch.float32 or q.dtype != torch.float32:
            raise RuntimeError("mlp_classify (HIP) computes in fp32; got %s" % v.dtype)
        L, B, d = v.shape
        if L != 3 or tuple(q.shape) != (3, B, d):
            raise RuntimeError("v and q must both be [3, B, d]; got %s and %s" % (tuple(v.shape), tuple(q.shape)))
        mlp, K = W_s.shape[0], W_h.shape[0]
        if tuple(W_w.shape) != (d, d) or tuple(W_p.shape) != (d, 2 * d) or tuple(W_s.shape) != (mlp, 2 * d) \
                or tuple(W_h.shape) != (K, mlp):
            raise RuntimeError("MLPClassifier weights must be W_w [d,d], W_p [d,2d], W_s [mlp,2d], W_h [K,mlp]")
        lib = _lib.load()
        v, q = v.contiguous(), q.contiguous()
        ps = [t.contiguous() for t in (W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h)]
        need_grad = any(ctx.needs_input_grad)
        sb, fb, _ = _mlp_bytes(B, d, mlp, K)
        dev = v.device
        logits = torch.empty((B, K), device=dev, dtype=torch.float32)
        saved = torch.empty(sb // 4, device=dev, dtype=torch.float32) if need_grad else None
        ws = torch.empty(fb // 4, device=dev, dtype=torch.float32)
        p = _lib.MlpParams(*[t.data_ptr() for t in ps])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.coattn_mlp_forward(_ptr(v), _ptr(q), C.byref(p), _ptr(logits), _ptr(saved), _ptr(ws),
                                              B, d, mlp, K, _lib.F32, 0, stream), "coattn_mlp_forward")
        if need_grad:
            ctx.save_for_backward(v, q, saved, *ps)
            ctx.dims = (B, d, mlp, K)
        return logits

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        v, q, saved, *ps = ctx.saved_tensors
        B, d, mlp, K = ctx.dims
        lib = _lib.load()
        dev = v.device
        g = g.contiguous().float()
        _, _, bb = _mlp_bytes(B, d, mlp, K)
        ws = torch.empty(bb // 4, device=dev, dtype=torch.float32)
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        g_vq = torch.empty_like(v) if need_x else None
        grads = [torch.empty_like(t) for t in ps]
        p = _lib.MlpParams(*[t.data_ptr() for t in ps])
        pg = _lib.MlpParamGrads(*[t.data_ptr() for t in grads])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.coattn_mlp_backward(_ptr(v), _ptr(q), C.byref(p), _ptr(saved), _ptr(g), _ptr(g_vq),
                                               C.byref(pg), 0, _ptr(ws), B, d, mlp, K, _lib.F32, 0, stream),
                       "coattn_mlp_backward")
        return (g_vq if ctx.needs_input_grad[0] else None, g_vq if ctx.needs_input_grad[1] else None, *grads)


def mlp_classify(v: torch.Tensor, q: torch.Tensor, W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h) -> torch.Tensor:
    """v, q [3, B, d] (word, phrase, sentence) -> logits [B, K]; weights in nn.Linear layout."""
    return _MlpFn.apply(v, q, W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h)


def as_level_stack(feats: Sequence[torch.Tensor]) -> torch.Tensor:
    """The list of three [B, d] tensors the co-attention returns -> one [3, B, d] tensor.  When they are the
    row views of one [3, B, d] buffer (what ``ParallelCoAttention.forward`` hands out) that buffer itself is
    returned (no copy, and the gradient flows straight into it); anything else is stacked."""
    a, b, c = feats
    base = a._base
    if (base is not None and b._base is base and c._base is base and base.dim() == 3 and base.shape[0] == 3
            and base.is_contiguous() and tuple(base.shape[1:]) == tuple(a.shape)
            and a.data_ptr() == base.data_ptr() and b.data_ptr() == base[1].data_ptr()
            and c.data_ptr() == base[2].data_ptr()):
        return base
    return torch.stack([a, b, c])


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
        with torch.cuda.device(dev):
            _lib.check(lib.coattn_ce_forward(_ptr(z), _ptr(lab), _ptr(loss), _ptr(dz), _ptr(ws), B, K, _lib.F32,
                                             stream), "coattn_ce_forward")
        if need:
            ctx.save_for_backward(dz)
        return loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


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
