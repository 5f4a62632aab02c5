"""Developer probe: where a workgroup of the fused affinity+softmax+reduce forward kernel spends its time.

Needs a DIAGNOSTIC build of the library (never the shipped one):
    hipcc ... -DCOATTN_STAMPS=1 -c coattn_fused.hip ; link as tools/ab/libcoattn_stamps.so
    COATTN_LIB_PATH=$PWD/tools/ab/libcoattn_stamps.so python tools/probe_stamps.py
In that build wave 0 of every workgroup writes the 100 MHz constant clock (s_memrealtime) at its phase
boundaries into the forward workspace tail, which no other code reads.  Prints per-phase durations
(mean / min / max over workgroups, microseconds) and the spread of start / end times.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd import _lib  # noqa: E402
from bench import synth_features  # noqa: E402

B, N, T, d, L = (int(os.environ.get(k, v)) for k, v in (("B", 160), ("N", 196), ("T", 26), ("D", 512), ("L", 3)))
dev = torch.device("cuda", 0)
lib = _lib.load()
torch.manual_seed(0)
co = vqa_amd.ParallelCoAttention(d).to(dev)
ps = [t.detach().contiguous() for t in (co.W_v.weight, co.W_v.bias, co.W_q.weight, co.W_q.bias, co.w_v.weight,
                                       co.w_v.bias, co.w_q.weight, co.w_q.bias)]
sb, fb, _ = _lib.workspace_bytes(B, N, T, d, L)
p = _lib.Params(*[t.data_ptr() for t in ps])
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flags = _lib.FLAG_FAST16 if os.environ.get("PRECISION", "exact") == "fast" else 0
SETS = int(os.environ.get("SETS", "1"))          # > 1: launches rotate over this many independent buffer sets (cache-cold operands)
sets = []
for k in range(SETS):
    V, Qs = synth_features(B, N, T, d, dev, seed=77 + k, L=L)
    saved = torch.empty(sb // 4, device=dev)
    ws = torch.zeros(fb // 4, device=dev)
    v = torch.empty(L, B, d, device=dev)
    q = torch.empty(L, B, d, device=dev)
    qptr = (C.c_void_p * L)(*[t.data_ptr() for t in Qs])
    vstr = (d * N, 1, N)
    if os.environ.get("LAYOUT", "lm") == "lm":
        V, vstr = V.permute(0, 2, 1).contiguous(), (N * d, d, 1)
    args = (V.data_ptr(), *vstr, qptr, C.byref(p), v.data_ptr(), q.data_ptr(), saved.data_ptr(), ws.data_ptr(),
            B, N, T, d, L, _lib.F32, flags, stream)
    _lib.check(lib.coattn_forward(*args), "coattn_forward")
    sets.append((args, ws, (V, Qs, saved, v, q, qptr)))
for i in range(int(os.environ.get("WARM", "300")) // SETS * SETS):   # clocks ramp over the first ~30 ms of load: read the stamps warm
    _lib.check(lib.coattn_attention_forward(*sets[i % SETS][0]), "coattn_attention_forward")
torch.cuda.synchronize()
ws = sets[-1][1]                                  # (the set of the last launch)
print("precision %s, %d buffer set(s)" % ("fast16" if flags else "exact", SETS))
nblk = ((B + 7) // 8) * L * 8
tail = ws[sb // 4:].view(torch.int64)[: nblk * 64].cpu().numpy().reshape(nblk, 64).astype(np.float64)
live = tail[:, 0] > 0
if not live.any():
    sys.exit("no stamps found: this is not a -DCOATTN_STAMPS=1 build")
st = tail[live] * 0.01                       # 100 MHz ticks -> microseconds
t0 = st[:, 0].min()
names = ["phase 1 MFMA loop (A = Q V^T)", "cross-wave sum + C = tanh", "phase 2 tile loop (H_v scores, H_q)",
         "H_q epilogue / barrier", "softmaxes + q"]
print("workgroups with stamps: %d; kernel span (first start -> last end): %.1f us" % (live.sum(), st[:, 5].max() - t0))
print("start spread: %.1f us; end spread: %.1f us" % (st[:, 0].max() - t0, st[:, 5].max() - st[:, 5].min()))
for k, nm in enumerate(names):
    dt = st[:, k + 1] - st[:, k]
    print("  %-40s mean %6.1f  min %6.1f  max %6.1f us" % (nm, dt.mean(), dt.min(), dt.max()))
tot = st[:, 5] - st[:, 0]
print("  %-40s mean %6.1f  min %6.1f  max %6.1f us" % ("whole workgroup", tot.mean(), tot.min(), tot.max()))
for k in range(6):
    print("  boundary %d reached at (rel. to first start): mean %6.1f  min %6.1f  max %6.1f us"
          % (k, (st[:, k] - t0).mean(), (st[:, k] - t0).min(), (st[:, k] - t0).max()))

if st.shape[1] > 6 and (tail[live][:, 6] > 0).all():
    print("  of the cross-wave sum + tanh span: tree %.1f us, tanh / split / image %.1f us"
          % ((st[:, 6] - st[:, 1]).mean(), (st[:, 2] - st[:, 6]).mean()))

# (the per-tile stamps of the removed tile-pipelined kernel are no longer produced)
nt = (N + 15) // 16
if st.shape[1] >= 8 + 4 * nt and (st[:, 8:8 + 4 * nt] > 0).all():
    tl = st[:, 8:8 + 4 * nt].reshape(-1, nt, 4)
    steps = tl[:, :, 1] - tl[:, :, 0]
    bar = tl[:, :, 2] - tl[:, :, 1]
    fin = tl[:, :, 3] - tl[:, :, 2]
    print("per tile (mean over workgroups and tiles): 4 MFMA steps %.2f us, partial write + barrier %.2f us, "
          "sum/tanh/C store %.2f us" % (steps.mean(), bar.mean(), fin.mean()))
    lone = (st[:, 5] - st[:, 0]) < np.percentile(st[:, 5] - st[:, 0], 8)
    print("  workgroups alone on their CU (fastest 8%%): steps %.2f, barrier %.2f, finish %.2f us"
          % (steps[lone].mean(), bar[lone].mean(), fin[lone].mean()))
    print("  steps per tile index:", " ".join("%.1f" % x for x in steps.mean(axis=0)))

# shader-cycle stamps (s_memtime) around the tile loop of coattn_fused.hip: cycles per tile and the clock they imply
if st.shape[1] > 7 and (tail[live][:, 7] > 0).all() and (tail[live][:, 6] > 0).all() and False:
    cyc = tail[live][:, 7] - tail[live][:, 6]
    us = st[:, 3] - st[:, 2]
    print("tile loop: %.0f shader cycles per workgroup (%.0f per tile) in %.1f us -> %.2f GHz; MFMA issue cycles per wave: %d"
          % (cyc.mean(), cyc.mean() / nt, us.mean(), (cyc / us).mean() / 1e3, nt * 120 * 32))

# per-pass stamps of coattn_fwd32.hip: pass start, tile loop done, epilogue done (two passes per 128-channel slice)
npass = 2 * max(1, d // (128 * (4 if d % 512 == 0 else 2)))
if st.shape[1] >= 8 + 3 * npass and (st[:, 8:8 + 3 * npass] > 0).all():
    ps = st[:, 8:8 + 3 * npass].reshape(-1, npass, 3)
    print("per pass (mean over workgroups): " + "; ".join(
        "pass %d: tile loop %.1f us, epilogue %.1f us" % (i, (ps[:, i, 1] - ps[:, i, 0]).mean(), (ps[:, i, 2] - ps[:, i, 1]).mean())
        for i in range(npass)))
