"""GPU-side test helper: run the C-ABI forward/backward directly (ctypes) on CUDA tensors and
return every output plus the saved forward state, parsed with the layout of csrc/api.hip."""
import ctypes as C

import torch

import vqa_amd
from vqa_amd import _lib

IMPL = {"general": _lib.IMPL_GENERAL, "fused": _lib.IMPL_FUSED, "auto": _lib.IMPL_AUTO}


def _al64(n):
    return (n + 63) & ~63


def saved_views(saved, B, N, T, d, L):
    # (P_v / P_q: on the fused path they are stored multiplied by 2 log2(e) -- csrc/fused.h, kPScale; tests do not
    #  compare them with the oracle)
    o = 0
    out = {}
    for name, shape in (("P_v", (B, N, d)), ("P_q", (L, B, T, d)), ("C", (L, B, T, N)), ("a_v", (L, B, N)),
                        ("a_q", (L, B, T)), ("H_q", (L, B, T, d))):
        n = 1
        for s in shape:
            n *= s
        out[name] = saved[o:o + n].view(*shape)
        o += _al64(n)
    return out


last_status = None
MODES = (False, True)       # exact3 of run_hip: the tolerance mode (COATTN_FLAG_FAST16) / flags = 0, fp32-accurate products
MODE_IDS = ("fast16", "exact")
LAYOUTS = ("cm", "lm")      # channel-major [B,d,N] (the reference's NCHW encoder) / location-major [B,N,d] (channels_last)


def run_hip(V, Qs, P, gv=None, gq=None, impl="general", need_dv=True, accumulate=0, grads_init=None, bf16_proj=False,
            layout="cm", exact3=False):
    """V [B,d,N] (channel-major values; `layout` selects the PHYSICAL layout handed to the C-ABI: "cm" as is, "lm" a
    [B,N,d] buffer), Qs list of [B,T,d], P dict of reference-named params (CPU or CUDA tensors).
    Returns dict with v,q and saved state; with gv/gq also all gradients (dV_phys always as [B,d,N] values)."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    V = V.to(dev).contiguous()
    B, d, N = V.shape
    if layout == "lm":
        Vbuf = V.permute(0, 2, 1).contiguous()          # [B,N,d]
        vstr = (N * d, d, 1)
    else:
        Vbuf = V
        vstr = (d * N, 1, N)
    Qs = [q.to(dev).contiguous() for q in Qs]
    names = ("W_v.weight", "W_v.bias", "W_q.weight", "W_q.bias", "w_v.weight", "w_v.bias", "w_q.weight", "w_q.bias")
    ps = [P[k].to(dev).contiguous() for k in names]
    assert gv is None or gv.shape[0] == len(Qs)
    T = Qs[0].shape[1]
    L = len(Qs)
    # (the tolerance mode -- what train.Trainer runs -- unless exact3: flags = 0, the C-ABI's default, fp32-accurate products)
    flag = IMPL[impl] | (_lib.FLAG_BF16_PROJ if bf16_proj else 0) | (0 if (exact3 or bf16_proj) else _lib.FLAG_FAST16)
    sb, fb, bb = _lib.workspace_bytes(B, N, T, d, L, flag)
    v = torch.full((L, B, d), float("nan"), device=dev)
    q = torch.full((L, B, d), float("nan"), device=dev)
    saved = torch.full((sb // 4,), float("nan"), device=dev)
    ws = torch.full((fb // 4,), float("nan"), device=dev)
    qptr = (C.c_void_p * L)(*[t.data_ptr() for t in Qs])
    p = _lib.Params(*[t.data_ptr() for t in ps])
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.coattn_forward(Vbuf.data_ptr(), *vstr, qptr, C.byref(p), v.data_ptr(), q.data_ptr(), saved.data_ptr(),
                                  ws.data_ptr(), B, N, T, d, L, _lib.F32, flag, C.c_void_p(stream)), "coattn_forward")
    torch.cuda.synchronize()
    out = {"v": v, "q": q}
    amax = (C.c_float * 2)()
    global last_status                      # (rc, largest out-of-range |activation|, largest |256 W|) of this forward call
    last_status = (lib.coattn_status(saved.data_ptr(), B, N, T, d, L, _lib.F32, C.c_void_p(stream), amax), amax[0], amax[1])
    out.update(saved_views(saved, B, N, T, d, L))
    if gv is None:
        return out
    gv = gv.to(dev).contiguous()
    gq = gq.to(dev).contiguous()
    ws2 = torch.full((bb // 4,), float("nan"), device=dev)
    dV = torch.full_like(Vbuf, float("nan")) if need_dv else None
    dQs = [torch.full_like(t, float("nan")) for t in Qs]
    if grads_init is None:
        grads = [torch.full_like(t, float("nan")) for t in ps]
    else:
        grads = [g.to(dev).clone() for g in grads_init]
    pg = _lib.ParamGrads(*[t.data_ptr() for t in grads])
    dqptr = (C.c_void_p * L)(*[t.data_ptr() for t in dQs])
    _lib.check(lib.coattn_backward(Vbuf.data_ptr(), *vstr, qptr, C.byref(p), saved.data_ptr(), gv.data_ptr(),
                                   gq.data_ptr(), dV.data_ptr() if need_dv else None, *vstr, dqptr, C.byref(pg),
                                   accumulate, ws2.data_ptr(), B, N, T, d, L, _lib.F32, flag, C.c_void_p(stream)),
               "coattn_backward")
    torch.cuda.synchronize()
    out["dV_phys"] = (dV.permute(0, 2, 1).contiguous() if layout == "lm" else dV) if need_dv else None
    out["dQ"] = torch.stack(dQs)
    for k, g in zip(names, grads):
        out["d" + k] = g
    return out
