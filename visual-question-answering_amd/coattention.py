"""``ParallelCoAttention``: drop-in for /root/reference/model.py:337-397 on MI355X.

Same constructor, attribute names (=> identical ``state_dict`` keys ``W_b, W_v, W_q, w_v, w_q``)
and ``forward(x_img, x_ques_hierarchy)`` signature as the reference class; the computation runs
in the HIP library through the C-ABI of ``include/coattn.h`` on the caller's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib


def _impl_flag() -> int:
    f = {"auto": _lib.IMPL_AUTO, "general": _lib.IMPL_GENERAL, "fused": _lib.IMPL_FUSED}[
        os.environ.get("COATTN_IMPL", "auto")]
    if os.environ.get("COATTN_BF16_PROJ", "0") not in ("0", ""):
        f |= _lib.FLAG_BF16_PROJ
    return f


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _is_native(x_img: torch.Tensor, lm_only: bool = False) -> bool:
    B, N, d = x_img.shape
    sB, sN, sD = _strides(x_img)
    ext = (N - 1) * sN + (d - 1) * sD
    lm = sD == 1 and sN == d
    cm = sN == 1 and sD == N and N % 4 == 0 and not lm_only
    return bool((lm or cm) and sB > ext and x_img.data_ptr() % 16 == 0)


# Frozen channel-major features (the reference's NCHW view at N = 196): the kernels read them in place through the C-ABI's
# strides, but the location-major kernels are faster by more than the library's one-pass conversion costs (cfg 2, N = 196: the
# isolated hot path 0.71 ms in place, 0.64 location-major, the conversion 28 us) -- so the host converts.  "inplace" keeps
# them where they lie (VQA_CM_FEATURES=inplace, or set this attribute).
CM_FEATURES = os.environ.get("VQA_CM_FEATURES", "convert")


def _native_layout(x_img: torch.Tensor) -> torch.Tensor:
    """The fp32 module input x_img[B,N,d] as the kernels take it: by pointer + element strides, no copy, when it is
    location-major (a channels_last encoder: contiguous [B,N,d]) or channel-major (the permuted view of an NCHW
    encoder, model.py:215-217: strides (d N, 1, N)) with rows that are 16-byte multiples (N % 4 == 0: N = 196);
    anything else is made contiguous -- location-major -- once.  That includes channel-major features at N = 49: their
    196-byte rows push the projection GEMMs onto dword loads (2 x 80 us per step), a 16 MB re-layout costs 8 us.
    (Inside autograd, where the copy must stay differentiable; features that need no gradient come through
    `native_features`, whose copy is the library's one-pass kernel.)"""
    if _is_native(x_img):
        return x_img
    return x_img.contiguous()


def native_features(x_img: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """Image features that need NO gradient (the frozen encoder of every BASELINE config, model.py:239-241), as the
    encoder left them -- fp32 or bf16 (an autocast encoder, main.py:73, :185), any strides -- in a layout the kernels
    run on: the tensor itself when it is fp32 and native already (see `_native_layout`), else ONE pass of
    coattn_features_native into fp32 [B,N,d] (`out`, if given: a contiguous fp32 buffer of that shape).  What torch
    makes of the same input is an up-cast and a strided copy: two passes, the second at a quarter of the memory rate."""
    if x_img.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("native_features is for features without a gradient (use the module: its copy is differentiable)")
    if not x_img.is_cuda or x_img.dtype not in (torch.float32, torch.bfloat16):
        x = x_img.detach().to(torch.float32)
        x = _native_layout(x)
        if out is None:
            return x
        out.copy_(x)
        return out
    if out is None and x_img.dtype == torch.float32 and _is_native(x_img, lm_only=CM_FEATURES != "inplace"):
        return x_img
    B, N, d = x_img.shape
    if _strides(x_img)[2] == 1:                      # rows along the channels already: a plain (vectorised) up-cast / copy
        if out is None:
            return _native_layout(x_img.to(torch.float32))
        out.copy_(x_img)
        return out
    if out is None:
        out = torch.empty((B, N, d), device=x_img.device, dtype=torch.float32)
    elif (tuple(out.shape) != (B, N, d) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x_img.device):
        raise RuntimeError("native_features: `out` must be a contiguous fp32 [B,N,d] tensor on the features' device")
    lib = _lib.load()
    sB, sN, sD = _strides(x_img)
    if min(sB, sN, sD) < 0:
        raise RuntimeError("native_features: negative strides")
    with _lib.on_device(x_img.device):
        _lib.check(lib.coattn_features_native(_ptr(x_img), _lib.BF16 if x_img.dtype == torch.bfloat16 else _lib.F32,
                                              sB, sN, sD, _ptr(out), B, N, d,
                                              C.c_void_p(torch.cuda.current_stream(x_img.device).cuda_stream)),
                   "coattn_features_native")
    return out


def _strides(x: torch.Tensor):
    """Element strides of x[B,N,d] for the C-ABI.  The stride torch reports for a size-1 dimension is arbitrary (and
    ``.contiguous()`` keeps it): such dimensions get the stride a contiguous [B,N,d] tensor would have."""
    B, N, d = x.shape
    sB, sN, sD = x.stride()
    if d == 1:
        sD = 1
    if N == 1:
        sN = d * sD if sD == 1 else 1
    if B == 1:
        sB = max(sB, (N - 1) * sN + (d - 1) * sD + 1, N * d)
    return sB, sN, sD


class _CoAttentionFn(torch.autograd.Function):
    """forward -> coattn_forward, backward -> coattn_backward (autograd of model.py:372-392)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # fp32 island under autocast
    def forward(ctx, x_img, W_v, b_v, W_q, b_q, w_v, c_v, w_q, c_q, impl, *x_ques):
        if not x_img.is_cuda:
            raise RuntimeError("ParallelCoAttention (HIP) needs tensors on the GPU; there is no CPU fallback")
        if x_img.dtype != torch.float32 or any(q.dtype != torch.float32 for q in x_ques):
            raise RuntimeError("ParallelCoAttention (HIP) computes in fp32; got %s" % x_img.dtype)
        lib = _lib.load()
        L = len(x_ques)
        B, N, d = x_img.shape
        T = x_ques[0].shape[1]
        for q in x_ques:
            if tuple(q.shape) != (B, T, d):
                raise RuntimeError("question features must all be [B,T,d] = %s, got %s" % ((B, T, d), tuple(q.shape)))
        V = _native_layout(x_img)
        Qs = [q.contiguous() for q in x_ques]
        params = [t.contiguous() for t in (W_v, b_v, W_q, b_q, w_v, c_v, w_q, c_q)]
        need_grad = any(ctx.needs_input_grad)
        sb, fb, _ = _lib.workspace_bytes(B, N, T, d, L, impl)
        dev = x_img.device
        out_v = torch.empty((L, B, d), device=dev, dtype=torch.float32)
        out_q = torch.empty((L, B, d), device=dev, dtype=torch.float32)
        saved = torch.empty(sb // 4, device=dev, dtype=torch.float32) if need_grad else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _lib.scratch(fb, dev, stream)            # per-stream scratch, reused from call to call
        qptr = (C.c_void_p * L)(*[q.data_ptr() for q in Qs])
        p = _lib.Params(*[t.data_ptr() for t in params])
        with _lib.on_device(dev):
            _lib.check(lib.coattn_forward(_ptr(V), *_strides(V), qptr, C.byref(p), _ptr(out_v), _ptr(out_q), _ptr(saved),
                                          _ptr(ws), B, N, T, d, L, _lib.F32, impl, C.c_void_p(stream)),
                       "coattn_forward")
        if impl & _lib.FLAG_FAST16:                   # tolerance mode: where _lib.check_range() finds this call's status words
            _lib.note_status("coattn", saved if saved is not None else ws, (B, N, T, d, L), dev)
        if need_grad:
            ctx.save_for_backward(V, saved, *params, *Qs)
            ctx.dims = (B, N, T, d, L, impl)
        return out_v, out_q

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_v, g_q):
        lib = _lib.load()
        B, N, T, d, L, impl = ctx.dims
        sv = ctx.saved_tensors
        V, saved, params, Qs = sv[0], sv[1], sv[2:10], sv[10:]
        dev = V.device
        g_v = g_v.contiguous() if g_v is not None else torch.zeros((L, B, d), device=dev)
        g_q = g_q.contiguous() if g_q is not None else torch.zeros((L, B, d), device=dev)
        _, _, bb = _lib.workspace_bytes(B, N, T, d, L, impl)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ws = _lib.scratch(bb, dev, stream)
        need_dv = ctx.needs_input_grad[0]
        dV = None
        if need_dv:                         # gradient of x_img[B,N,d] in the layout of x_img itself
            dV = (torch.empty((B, N, d), device=dev) if V.stride(2) == 1
                  else torch.empty((B, d, N), device=dev).permute(0, 2, 1))
        dQs = [torch.empty_like(q) for q in Qs]
        grads = [torch.empty_like(t) for t in params]
        pg = _lib.ParamGrads(*[t.data_ptr() for t in grads])
        p = _lib.Params(*[t.data_ptr() for t in params])
        qptr = (C.c_void_p * L)(*[q.data_ptr() for q in Qs])
        dqptr = (C.c_void_p * L)(*[q.data_ptr() for q in dQs])
        with _lib.on_device(dev):
            _lib.check(lib.coattn_backward(_ptr(V), *_strides(V), qptr, C.byref(p), _ptr(saved), _ptr(g_v), _ptr(g_q),
                                           _ptr(dV), *(_strides(dV) if need_dv else (0, 0, 0)), dqptr, C.byref(pg), 0,
                                           _ptr(ws), B, N, T, d, L, _lib.F32, impl, C.c_void_p(stream)),
                       "coattn_backward")
        return (dV, *grads, None, *dQs)


def coattention(x_img: torch.Tensor, x_ques: Sequence[torch.Tensor], W_v, b_v, W_q, b_q, w_v, c_v, w_q, c_q,
                impl: int | None = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Functional form: returns (v, q), each [L,B,d]."""
    if impl is None:
        impl = _impl_flag()
    return _CoAttentionFn.apply(x_img, W_v, b_v, W_q, b_q, w_v, c_v, w_q, c_q, impl, *x_ques)


class ParallelCoAttention(nn.Module):
    """Parallel co-attention over image and word/phrase/sentence question features.

    Mirrors the reference class (model.py:337-397): ``W_b`` is constructed but never used
    (model.py:347 vs :377) so it stays in the state_dict and never receives a gradient;
    W_v/W_q/w_v/w_q are ``nn.Linear`` with biases; one weight set serves all levels; the
    softmax over question positions is unmasked (model.py:388).
    """

    def __init__(self, hidden_dim: int):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.W_b = nn.Linear(hidden_dim, hidden_dim)     # dead in the reference's forward
        self.W_v = nn.Linear(hidden_dim, hidden_dim)
        self.W_q = nn.Linear(hidden_dim, hidden_dim)
        self.w_v = nn.Linear(hidden_dim, 1)
        self.w_q = nn.Linear(hidden_dim, 1)
        # reduced-precision mode (apex O1 analogue, main.py:185): projections on the bf16 MFMA
        self.bf16_projections = False
        # Precision of the fp32 mode (include/coattn.h "Widths of the fp32 mode").  False -- the default, as the reference:
        # every product fp32-accurate over fp32's range.  True -- the tolerance mode (forward products on two FP16 pieces =
        # 22 significand bits, backward on two bf16 pieces = 16; inside the 1e-4 contract for operands below 65,504 in
        # magnitude; `vqa_amd.check_range()` reports an operand that was not): what train.Trainer(precision="fast") sets.
        self.fast_products = _lib.default_fast()

    def forward(self, x_img: torch.Tensor, x_ques_hierarchy: Sequence[torch.Tensor]) -> Tuple[List, List]:
        """x_img [B,N,d]; x_ques_hierarchy: list of [B,T,d] -> (list of v_l [B,d], list of q_l [B,d])."""
        impl = _impl_flag() | (_lib.FLAG_BF16_PROJ if self.bf16_projections else 0) | _lib.precision_flag(self.fast_products)
        if x_img.is_cuda and not (x_img.requires_grad and torch.is_grad_enabled()):
            x_img = native_features(x_img)           # frozen encoder: bf16 / non-native strides in one library pass
        v, q = coattention(x_img, list(x_ques_hierarchy), self.W_v.weight, self.W_v.bias, self.W_q.weight,
                           self.W_q.bias, self.w_v.weight, self.w_v.bias, self.w_q.weight, self.w_q.bias, impl=impl)
        n = v.shape[0]
        return [v[l] for l in range(n)], [q[l] for l in range(n)]
