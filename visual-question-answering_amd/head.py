"""``answer_head``: the reference's ``MLPClassifier`` (model.py:400-434) and the ``nn.CrossEntropyLoss()`` of the
train step (main.py:94, :214) with their backward, through ``coattn_head_forward`` / ``coattn_head_backward`` of
``include/coattn.h`` (``csrc/head.hip``) on the caller's current stream.  SURVEY.md section 8f-1.

The q_l + v_l adds, the concatenations, bias + tanh and tanh' live in the operand addressing / epilogues of the
tile products; with labels the loss and d loss / d logits come out of the same forward call.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib

_ws_cache = {}
_last = None          # (saved, B, d, mlp, K, device) of the last forward with labels: check_labels() reads its status word


def _workspace_bytes(B, d, mlp, K):
    key = (B, d, mlp, K)
    hit = _ws_cache.get(key)
    if hit is None:
        s, w = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().coattn_head_workspace_bytes(B, d, mlp, K, _lib.F32, C.byref(s), C.byref(w)),
                   "coattn_head_workspace_bytes")
        hit = _ws_cache[key] = (s.value, w.value)
    return hit


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _flags() -> int:
    """VQA_HEAD_PERSISTENT=1 (developer switch): one launch per direction with grid-wide barriers between the layers."""
    return 1 if os.environ.get("VQA_HEAD_PERSISTENT", "0") not in ("0", "") else 0


def _rows(t):
    """Host array of the three [B,d] row-block pointers of a contiguous [3,B,d] tensor."""
    step = t.stride(0) * t.element_size()
    return (C.c_void_p * 3)(*[t.data_ptr() + l * step for l in range(3)])


class _HeadFn(torch.autograd.Function):
    """(v [3,B,d], q [3,B,d], 8 parameters, labels or None) -> (logits [B,K], loss [] or None)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # fp32 island under autocast
    def forward(ctx, v, q, W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h, labels, bf16=False):
        ctx.flags = _flags() | (_lib.FLAG_BF16_PROJ if bf16 else 0)
        if not v.is_cuda:
            raise RuntimeError("answer_head (HIP) needs tensors on the GPU; there is no CPU fallback")
        if v.dtype != torch.float32 or q.dtype != torch.float32:
            raise RuntimeError("answer_head (HIP) computes in fp32; got %s" % v.dtype)
        if v.dim() != 3 or v.shape[0] != 3 or v.shape != q.shape:
            raise RuntimeError("answer_head: v and q must both be [3,B,d], got %s / %s" % (tuple(v.shape), tuple(q.shape)))
        _, B, d = v.shape
        mlp, K = W_s.shape[0], W_h.shape[0]
        if (tuple(W_w.shape) != (d, d) or tuple(W_p.shape) != (d, 2 * d) or tuple(W_s.shape) != (mlp, 2 * d)
                or tuple(W_h.shape) != (K, mlp)):
            raise RuntimeError("answer_head: weight shapes do not match MLPClassifier(hidden_dim=%d, mlp_dim, K)" % d)
        if labels is not None and (labels.dtype != torch.int64 or tuple(labels.shape) != (B,)):
            raise RuntimeError("answer_head: labels must be int64 [B]")
        lib = _lib.load()
        v, q = v.contiguous(), q.contiguous()
        ps = [t.contiguous() for t in (W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h)]
        dev = v.device
        sb, _ = _workspace_bytes(B, d, mlp, K)
        saved = torch.empty(sb // 4, device=dev, dtype=torch.float32)
        logits = torch.empty((B, K), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32) if labels is not None else None
        lab = labels.contiguous() if labels is not None else None
        p = _lib.HeadParams(*[t.data_ptr() for t in ps])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with _lib.on_device(dev):
            _lib.check(lib.coattn_head_forward(_rows(v), _rows(q), C.byref(p), _ptr(lab), _ptr(logits), _ptr(loss),
                                               _ptr(saved), B, d, mlp, K, _lib.F32, ctx.flags, stream), "coattn_head_forward")
        if labels is not None:
            global _last
            _last = (saved, B, d, mlp, K, dev)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(v, q, saved, *ps)
            ctx.dims = (B, d, mlp, K)
            ctx.has_loss = labels is not None
        return logits, loss

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_logits, g_loss):
        v, q, saved, *ps = ctx.saved_tensors
        B, d, mlp, K = ctx.dims
        lib = _lib.load()
        dev = v.device
        if not ctx.has_loss:
            g_loss = None
        if g_logits is None and g_loss is None:
            return (None,) * 12
        g_logits = g_logits.contiguous().float() if g_logits is not None else None
        g_loss = g_loss.contiguous().float() if g_loss is not None else None
        _, wb = _workspace_bytes(B, d, mlp, K)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _lib.scratch(wb, dev, stream)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        dx = torch.empty_like(v) if need_in else None        # d(q_l + v_l): the gradient of BOTH v and q
        grads = [torch.empty_like(t) for t in ps]
        p = _lib.HeadParams(*[t.data_ptr() for t in ps])
        pg = _lib.HeadParamGrads(*[t.data_ptr() for t in grads])
        with _lib.on_device(dev):
            _lib.check(lib.coattn_head_backward(_rows(v), _rows(q), C.byref(p), _ptr(saved), _ptr(g_loss), _ptr(g_logits),
                                                _rows(dx) if need_in else None, None, C.byref(pg), 0, _ptr(ws),
                                                B, d, mlp, K, _lib.F32, ctx.flags, C.c_void_p(stream)), "coattn_head_backward")
        return (dx if ctx.needs_input_grad[0] else None, dx if ctx.needs_input_grad[1] else None, *grads, None, None)


def check_labels() -> None:
    """nn.CrossEntropyLoss raises on a label outside [0, K); the HIP head stays asynchronous, makes the loss NaN and
    sets a status word instead.  This SYNCHRONISES the current stream and raises IndexError if the last forward with
    labels met such a label -- call it where the host synchronises anyway (when the loss is read)."""
    if _last is None:
        return
    saved, B, d, mlp, K, dev = _last
    lib = _lib.load()
    with _lib.on_device(dev):
        rc = lib.coattn_head_status(_ptr(saved), B, d, mlp, K, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc == -2:
        raise IndexError(lib.coattn_last_error().decode())
    _lib.check(rc, "coattn_head_status")


def _as_3bd(x) -> torch.Tensor:
    """[3,B,d] tensor from either such a tensor or a sequence of three [B,d] tensors -- without a copy when the
    three are the rows of one contiguous [3,B,d] tensor (what ParallelCoAttention.forward returns)."""
    if torch.is_tensor(x):
        return x
    x = list(x)
    if len(x) != 3:
        raise RuntimeError("answer_head: three levels (word, phrase, sentence) expected, got %d" % len(x))
    base = getattr(x[0], "_base", None)
    if (base is not None and base.dim() == 3 and base.shape[0] == 3 and base.is_contiguous()
            and all(t._base is base and t.shape == base.shape[1:] and t.is_contiguous()
                    and t.data_ptr() == base.data_ptr() + l * base.stride(0) * base.element_size()
                    for l, t in enumerate(x))):
        return base
    return torch.stack(x)


def answer_head(v, q, W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h, labels: Optional[torch.Tensor] = None, bf16: bool = False):
    """v, q: [3,B,d] tensors or sequences of three [B,d] tensors (attended image / question features of the word,
    phrase and sentence levels).  Returns logits [B,K] -- and, with int64 labels [B], (logits, mean cross entropy).
    bf16: the reduced-precision mode (operands of the four products and of their gradients rounded to bf16, one bf16 MFMA
    where the exact head issues eight f32 ones; fp32 accumulation, biases, tanh and loss) -- the apex-O1 analogue."""
    logits, loss = _HeadFn.apply(_as_3bd(v), _as_3bd(q), W_w, b_w, W_p, b_p, W_s, b_s, W_h, b_h, labels, bf16)
    return logits if labels is None else (logits, loss)
