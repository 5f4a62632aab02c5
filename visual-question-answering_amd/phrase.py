"""``phrase_conv_pool``: the phrase level of the question hierarchy (reference model.py:301-334) on
MI355X -- the three n-gram Conv1d + Tanh and the MaxPool2d((1,3)) over groups of 3 consecutive
channels as one im2col GEMM + pooling epilogue per direction (``csrc/phrase.hip``), through the
C-ABI of ``include/coattn.h`` on the caller's current stream.  SURVEY.md section 8f-3.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _workspace_bytes(B, T, E):
    s, f, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
    _lib.check(_lib.load().coattn_phrase_workspace_bytes(B, T, E, _lib.F32, C.byref(s), C.byref(f), C.byref(b)),
               "coattn_phrase_workspace_bytes")
    return s.value, f.value, b.value


class _PhraseConvPoolFn(torch.autograd.Function):
    """forward -> coattn_phrase_forward, backward -> coattn_phrase_backward."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # fp32 island under autocast
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, flags):
        if not x.is_cuda:
            raise RuntimeError("phrase_conv_pool (HIP) needs tensors on the GPU")
        if x.dtype != torch.float32:
            raise RuntimeError("phrase_conv_pool (HIP) computes in fp32; got %s" % x.dtype)
        B, T, E = x.shape
        if tuple(W1.shape) != (E, E, 1) or tuple(W2.shape) != (E, E, 2) or tuple(W3.shape) != (E, E, 3):
            raise RuntimeError("n-gram conv weights must be [E,E,1], [E,E,2], [E,E,3] with E = %d" % E)
        lib = _lib.load()
        X = x.contiguous()
        ps = [t.contiguous() for t in (W1, b1, W2, b2, W3, b3)]
        need_grad = any(ctx.needs_input_grad)
        sb, fb, _ = _workspace_bytes(B, T, E)
        out = torch.empty_like(X)
        saved = torch.empty(sb, dtype=torch.uint8, device=x.device) if need_grad else None
        ws = torch.empty(fb, dtype=torch.uint8, device=x.device)
        p = _lib.PhraseParams(*[t.data_ptr() for t in ps])
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        with _lib.on_device(x.device):
            _lib.check(lib.coattn_phrase_forward(_ptr(X), C.byref(p), _ptr(out), _ptr(saved), _ptr(ws), B, T, E,
                                                 _lib.F32, flags, stream), "coattn_phrase_forward")
        if (flags & _lib.FLAG_FAST16) and saved is not None:   # tolerance mode: this call's status words, for _lib.check_range()
            _lib.note_status("phrase", saved, (B, T, E), x.device)
        if need_grad:
            ctx.flags = flags
            ctx.save_for_backward(X, out, saved, *ps)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        X, out, saved, *ps = ctx.saved_tensors
        B, T, E = X.shape
        lib = _lib.load()
        g = g.contiguous().float()
        _, _, bb = _workspace_bytes(B, T, E)
        ws = torch.empty(bb, dtype=torch.uint8, device=X.device)
        dX = torch.empty_like(X) if ctx.needs_input_grad[0] else None
        grads = [torch.empty_like(t) for t in ps]
        p = _lib.PhraseParams(*[t.data_ptr() for t in ps])
        pg = _lib.PhraseParamGrads(*[t.data_ptr() for t in grads])
        stream = C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream)
        with _lib.on_device(X.device):
            _lib.check(lib.coattn_phrase_backward(_ptr(X), C.byref(p), _ptr(out), _ptr(saved), _ptr(g), _ptr(dX),
                                                  C.byref(pg), 0, _ptr(ws), B, T, E, _lib.F32, ctx.flags, stream),
                       "coattn_phrase_backward")
        return (dX, *grads, None)


def phrase_conv_pool(x, W1, b1, W2, b2, W3, b3, bf16=None, fast=False):
    """x [B,T,E] -> [B,T,E]; weights in torch Conv1d layout ([E,E,k]) of the unigram / bigram / trigram convs.
    bf16: contract on the bf16 MFMA (fp32 storage / accumulation); default: when CUDA autocast is on --
    the stock Conv1d modules would run in reduced precision there too.  fast: the tolerance mode of the fp32 products
    (include/coattn.h COATTN_FLAG_FAST16; default: fp32-accurate products)."""
    if bf16 is None:
        bf16 = x.is_cuda and torch.is_autocast_enabled("cuda")
    return _PhraseConvPoolFn.apply(x, W1, b1, W2, b2, W3, b3,
                                   (_lib.FLAG_BF16_PROJ if bf16 else 0) | _lib.precision_flag(fast and not bf16))
