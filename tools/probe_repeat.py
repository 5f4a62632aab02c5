#!/usr/bin/env python3
"""Developer tool: coattn_forward + coattn_backward through the C-ABI, R times on the same inputs into fresh NaN-filled
buffers; every region of `saved` and of the backward workspace is compared with the first run bit for bit -- names the first
intermediate that is not repeatable (a race shows as a region that differs between runs).
usage: tools/probe_repeat.py [B N T d layout repeats [flags: fused|general|bf16|exact3 ...]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
B, N, T, d = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1024, 196, 26, 256))]
layout = sys.argv[5] if len(sys.argv) > 5 else "cm"
R = int(sys.argv[6]) if len(sys.argv) > 6 else 8
L = 3
lib = _lib.load(); dev = torch.device("cuda:0")
torch.manual_seed(3)
V = torch.randn(B, d, N, device=dev).clamp_min_(0)
Vbuf, vstr = (V, (d * N, 1, N)) if layout == "cm" else (V.permute(0, 2, 1).contiguous(), (N * d, d, 1))
Qs = [torch.randn(B, T, d, device=dev) * (2.0 / d) ** 0.5 for _ in range(L)]
ps = [torch.randn(d, d, device=dev) / d ** 0.5, torch.randn(d, device=dev) * 0.1, torch.randn(d, d, device=dev) / d ** 0.5,
      torch.randn(d, device=dev) * 0.1, torch.randn(1, d, device=dev) / d ** 0.5, torch.randn(1, device=dev),
      torch.randn(1, d, device=dev) / d ** 0.5, torch.randn(1, device=dev)]
gv = torch.randn(L, B, d, device=dev); gq = torch.randn(L, B, d, device=dev)
opts = sys.argv[7:]
flag = _lib.IMPL_GENERAL if "general" in opts else _lib.IMPL_FUSED
if "bf16" in opts: flag |= _lib.FLAG_BF16_PROJ
if "exact3" not in opts and "bf16" not in opts: flag |= _lib.FLAG_FAST16    # (default: the tolerance mode train.Trainer runs)
need_dv = "dv" in opts
sb, fb, bb = _lib.workspace_bytes(B, N, T, d, L, flag)
al = lambda n: (n + 63) & ~63
def regions(names_sizes):
    o, out = 0, {}
    for n, s in names_sizes:
        out[n] = (o, s); o += al(s)
    return out
SV = regions([("Pv", B*N*d), ("Pq", L*B*T*d), ("C", L*B*T*N), ("av", L*B*N), ("aq", L*B*T), ("Hq", L*B*T*d)])
WS = regions([("dsv", L*B*N), ("dsq", L*B*32), ("dPq", L*B*T*d), ("dPv", L*B*N*d), ("dA", L*B*T*N), ("dwv_part", L*B*d),
              ("dbv_part", L*B*d), ("dbq_part", L*B*d), ("dwq_part", L*B*d), ("dcs_part", L*B*2)])
qptr = (C.c_void_p * L)(*[t.data_ptr() for t in Qs]); p = _lib.Params(*[t.data_ptr() for t in ps])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("gpu uuid", getattr(torch.cuda.get_device_properties(0), "uuid", "?"), "B N T d", B, N, T, d, layout, opts)
ref = None
nbad = 0
for it in range(R):
    v = torch.full((L, B, d), float("nan"), device=dev); q = torch.full((L, B, d), float("nan"), device=dev)
    saved = torch.full((sb // 4,), float("nan"), device=dev); ws = torch.full((fb // 4,), float("nan"), device=dev)
    ws2 = torch.full((bb // 4,), float("nan"), device=dev)
    dQs = [torch.full_like(t, float("nan")) for t in Qs]; grads = [torch.full_like(t, float("nan")) for t in ps]
    pg = _lib.ParamGrads(*[t.data_ptr() for t in grads]); dqptr = (C.c_void_p * L)(*[t.data_ptr() for t in dQs])
    _lib.check(lib.coattn_forward(Vbuf.data_ptr(), *vstr, qptr, C.byref(p), v.data_ptr(), q.data_ptr(), saved.data_ptr(),
                                  ws.data_ptr(), B, N, T, d, L, _lib.F32, flag, st), "fwd")
    dV = torch.full_like(Vbuf, float("nan")) if need_dv else None
    _lib.check(lib.coattn_backward(Vbuf.data_ptr(), *vstr, qptr, C.byref(p), saved.data_ptr(), gv.data_ptr(), gq.data_ptr(),
                                   dV.data_ptr() if need_dv else None, *vstr, dqptr, C.byref(pg), 0, ws2.data_ptr(), B, N, T, d, L, _lib.F32, flag, st), "bwd")
    torch.cuda.synchronize()
    cur = {"v": v, "q": q, "dQ": torch.stack(dQs)}
    if need_dv: cur["dV"] = dV
    cur.update({"grad%d" % i: g for i, g in enumerate(grads)})
    cur.update({"saved." + n: saved[o:o + s] for n, (o, s) in SV.items()})
    if flag & 3 == _lib.IMPL_FUSED and not (flag & _lib.FLAG_BF16_PROJ):      # (the fp32 fused backward's workspace layout)
        cur.update({"ws." + n: ws2[o:o + s] for n, (o, s) in WS.items() if n != "dsv"})
    if ref is None:
        ref = {k: t.clone() for k, t in cur.items()}
        print("run 0: nonfinite per tensor:", {k: int((~torch.isfinite(t)).sum()) for k, t in cur.items() if (~torch.isfinite(t)).any()})
        continue
    bad = {}
    for k, t in cur.items():
        a, b = t.view(torch.int32), ref[k].view(torch.int32)
        n = int((a != b).sum())
        if n:
            idx = int((a != b).flatten().nonzero()[0])
            bad[k] = (n, idx, float(t.flatten()[idx]), float(ref[k].flatten()[idx]))
    if bad or it % 20 == 0 or it == R - 1:
        print("run %d: %s" % (it, "identical" if not bad else bad), flush=True)
    nbad += bool(bad)
print("runs with a difference:", nbad, "of", R - 1)
