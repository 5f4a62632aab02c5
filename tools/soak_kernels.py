"""Kernel soak (VERDICT r2 item 2 / ADVICE): is the "faulty box" of profiles/r02_faulty_box_suite.log a box, or a
timing-dependent hazard in this repo's hand-scheduled kernels?

One process, ITERS rounds.  Every round runs, interleaved,
  stock     torch fp32 matmul 31,360 x 512 x 512 held to float64 over ALL rows + an elementwise/row-sum kernel held to
            bitwise repeatability (nothing of this repo: if THIS fails the box is bad)
  gemm_f32  coattn_gemm_f32 at M = 50,052 (both A layouts)          -- tests/test_gpu_gemm.py::test_gemm_large_m
  gemm_w    coattn_linear_forward at M = 31,360, N = K = 512         -- ::test_linear_presplit_weight[31360-512-512]
  gemm_tn   coattn_linear_weight_grad at M = 31,360                  -- ::test_linear_weight_grad[31360-512-512]
  coattn    module forward + backward, three shapes                  -- test_gpu_edges.py::test_repeated_runs_are_bitwise_identical
  cfg2      coattention forward + backward at B=160, N=49, lm        -- test_gpu_parity.py::test_full_size_cfg2_properties[fused-lm-49]
  gemm_bf   the reduced-precision GEMMs of gemm_bf.hip (LDS-DMA weight image, counted waits) at config 4's size,
            M = 7,840, N = K = 2,048: coattn_linear_forward / coattn_linear_weight_grad with COATTN_FLAG_BF16_PROJ
  big_cm    the module's forward + backward at B = 640, N = 196, channel-major features (a busy chip: round 4's race of the
            forward kernel showed there) -- every 4th round, bitwise against round 0
  cfg4      the module's forward + backward in the reduced-precision mode at config 4's full size (B=160, N=49, d=2048,
            frozen image encoder): single-product fused kernels, gemm_bf.hip with bf16-stored gradients -- every 4th round,
            bf16 tolerance against float64, bitwise against round 0
each held (a) to a float64 reference computed ONCE with stock torch ops on the same GPU, over every element, and (b) to
bitwise equality with the first round.  The first mismatch of every check is printed with index, got, expected, the
number of bad elements and the rows they sit in.  Prints the GPU UUID and clocks; exit code 1 on any mismatch.

    python tools/soak_kernels.py [--iters 3000] [--log profiles/r03_soak_<uuid>.log]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--log", type=str, default="")
ap.add_argument("--budget-s", type=float, default=420.0, help="stop starting new rounds after this many seconds")
args = ap.parse_args()

dev = torch.device("cuda", 0)
lib = _lib.load()
props = torch.cuda.get_device_properties(0)
uuid = str(getattr(props, "uuid", "unknown"))
lines = []


def say(msg):
    print(msg, flush=True)
    lines.append(msg)


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        card = j[sorted(j)[0]]
        return {k: v for k, v in card.items() if "sclk" in k or "mclk" in k or "fclk" in k}
    except Exception as e:                                           # noqa: BLE001
        return {"error": str(e)[:80]}


say("soak_kernels: GPU %s uuid %s, %d CUs, torch %s, lib version %d" % (props.name, uuid, props.multi_processor_count,
                                                                      torch.__version__, lib.coattn_version()))
say("clocks at start: %s" % json.dumps(clocks()))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
failures = {}


def report(name, it, got, want, tol=None, first=None):
    """Compare `got` with the float64 reference `want` (relative to max|want|) and, bitwise, with the first round."""
    ok = True
    g = got.detach()
    if want is not None:
        scale = want.abs().max().clamp_min(1e-30)
        err = (g.double() - want).abs() / scale
        bad = err > tol
        nb = int(bad.sum())
        if nb or not bool(torch.isfinite(g).all()):
            ok = False
            idx = torch.nonzero(bad | ~torch.isfinite(g).reshape(bad.shape))
            i0 = tuple(int(x) for x in idx[0])
            rows = sorted({int(r[0]) for r in idx[:4096]})
            say("MISMATCH %s round %d: %d of %d elements off float64 by > %.1e (max %.3e); first at %s got %.9g expected %.9g; "
                "rows (first dim) %s%s" % (name, it, nb, g.numel(), tol, float(err.max()), i0, float(g[i0]), float(want[i0]),
                                           rows[:24], " ..." if len(rows) > 24 else ""))
    if first is not None and not torch.equal(g, first):
        ok = False
        d = torch.nonzero(g != first)
        i0 = tuple(int(x) for x in d[0])
        rows = sorted({int(r[0]) for r in d[:4096]})
        say("NOT REPEATABLE %s round %d: %d of %d elements differ from round 0; first at %s got %.9g, round 0 %.9g; rows %s%s"
            % (name, it, d.shape[0], g.numel(), i0, float(g[i0]), float(first[i0]), rows[:24], " ..." if len(rows) > 24 else ""))
    if not ok:
        failures[name] = failures.get(name, 0) + 1
    return ok


# ---- operands and float64 references (stock torch, once) --------------------------------------------------------
torch.manual_seed(0)
sa = torch.randn(31360, 512, device=dev); sw = torch.randn(512, 512, device=dev) / 512 ** 0.5
stock_ref = sa.double() @ sw.double()

torch.manual_seed(7)
gM, gN, gK = 50052, 256, 128
gA = {True: torch.randn(gK, gM, device=dev), False: torch.randn(gM, gK, device=dev)}
gW = torch.randn(gN, gK, device=dev) / 8; gb = torch.randn(gN, device=dev)
g_ref = {am: (gA[am].double().T if am else gA[am].double()) @ gW.double().T + gb.double() for am in (True, False)}

torch.manual_seed(11)
lM, lN, lK = 31360, 512, 512
lx = torch.randn(lM, lK, device=dev); lW = torch.randn(lN, lK, device=dev) / lK ** 0.5; lb = torch.randn(lN, device=dev)
l_ref = lx.double() @ lW.double().t() + lb.double()
l_wimg = torch.empty(lib.coattn_linear_workspace_bytes(lN, lK) // 4, device=dev)

torch.manual_seed(21)
tdy = torch.randn(lM, 512, device=dev) * 0.1; tx = torch.randn(lM, 512, device=dev)
t_ref = tdy.double().t() @ tx.double()
t_ws = torch.empty(lib.coattn_linear_wgrad_workspace_bytes(512, 512) // 4, device=dev)


torch.manual_seed(31)
bM, bd = 7840, 2048
bx = torch.randn(bM, bd, device=dev); bW = torch.randn(bd, bd, device=dev) / bd ** 0.5; bb = torch.randn(bd, device=dev)
bdy = torch.randn(bM, bd, device=dev) * 0.1
b_ref = bx.bfloat16().double() @ bW.bfloat16().double().t() + bb.double()
bt_ref = bdy.bfloat16().double().t() @ bx.bfloat16().double()
b_wimg = torch.empty(lib.coattn_linear_workspace_bytes(bd, bd) // 4, device=dev)
bt_ws = torch.empty(lib.coattn_linear_wgrad_workspace_bytes(bd, bd) // 4, device=dev)


def coattn_f64(x, Qs, co):
    """ParallelCoAttention.forward (reference model.py:372-392) in float64 with stock torch ops."""
    W_v, b_v, W_q, b_q = (t.double() for t in (co.W_v.weight, co.W_v.bias, co.W_q.weight, co.W_q.bias))
    w_v, c_v, w_q, c_q = (t.double() for t in (co.w_v.weight, co.w_v.bias, co.w_q.weight, co.w_q.bias))
    vs, qs = [], []
    Pv = x @ W_v.t() + b_v
    for Q in Qs:
        Cm = torch.tanh(Q @ x.transpose(1, 2))
        Pq = Q @ W_q.t() + b_q
        Hv = torch.tanh(Pv + Cm.transpose(1, 2) @ Pq)
        Hq = torch.tanh(Pq + Cm @ Pv)
        av = torch.softmax(Hv @ w_v.t() + c_v, dim=1)
        aq = torch.softmax(Hq @ w_q.t() + c_q, dim=1)
        vs.append((av * x).sum(1)); qs.append((aq * Q).sum(1))
    return vs, qs


class CoCase:
    def __init__(self, B, N, T, d, lay, seed, scale, bf16=False):
        """bf16: the reduced-precision mode with a frozen image encoder (no dV): single-product fused kernels, gemm_bf.hip,
        dP_v / dP_q stored as bf16 -- held to bf16 tolerance against float64 and to bitwise repeatability."""
        torch.manual_seed(seed)
        self.name = "coattn[B%d,N%d,T%d,d%d,%s%s]" % (B, N, T, d, lay, ",bf16" if bf16 else "")
        self.tol = 5e-2 if bf16 else 1e-4
        self.co = vqa_amd.ParallelCoAttention(d).to(dev)
        self.co.bf16_projections = bf16
        x = (torch.randn(B, d, N, device=dev) * scale).permute(0, 2, 1)
        self.x = (x.contiguous() if lay == "lm" else x).requires_grad_(not bf16)
        # questions as the reference's hierarchy leaves them: descending lengths, rows past a length exact zeros (round 6: the
        # exact mode finds those rows and leaves them out of P_q's projection and of dW_q -- bitmap, compacted tiles, LDS row maps)
        lens = torch.tensor(sorted([T] + [1 + (7 * i) % T for i in range(B - 1)], reverse=True), device=dev)
        mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).unsqueeze(-1).float()
        self.Qs = [(torch.randn(B, T, d, device=dev) * scale * mask).requires_grad_(True) for _ in range(3)]
        self.gv = torch.randn(3, B, d, device=dev); self.gq = torch.randn(3, B, d, device=dev)
        self.first = None
        # float64 reference through autograd of the stock ops
        xr = self.x.detach().double().requires_grad_(True)
        Qr = [q.detach().double().requires_grad_(True) for q in self.Qs]
        prm = [p for n, p in self.co.named_parameters() if not n.startswith("W_b")]
        prr = [p.detach().double().requires_grad_(True) for p in prm]

        class D:   # a stand-in carrying float64 leaves under the module's attribute names
            pass
        dd = D()
        it = iter(prr)
        for nm in ("W_v", "W_q", "w_v", "w_q"):
            o = D(); o.weight = next(it); o.bias = next(it); setattr(dd, nm, o)
        vs, qs = coattn_f64(xr, Qr, dd)
        loss = sum((vs[l] * self.gv[l].double()).sum() + (qs[l] * self.gq[l].double()).sum() for l in range(3))
        loss.backward()
        self.ref = [t.detach() for t in vs + qs] + [xr.grad if not bf16 else None] + [q.grad for q in Qr] + [p.grad for p in prr]
        self.prm = prm

    def run(self):
        for t in [self.x] + self.Qs + self.prm:
            t.grad = None
        v, q = self.co(self.x, self.Qs)
        torch.autograd.backward([torch.stack(v), torch.stack(q)], [self.gv, self.gq])
        return [t.detach() for t in v + q] + [self.x.grad] + [t.grad for t in self.Qs] + [p.grad for p in self.prm]   # (x.grad None when frozen)


cases = [CoCase(23, 196, 26, 512, "lm", 7, 0.5), CoCase(9, 49, 26, 512, "cm", 8, 0.5), CoCase(3, 100, 17, 1024, "lm", 9, 0.5),
         CoCase(160, 49, 26, 512, "lm", 10, (2.0 / 512) ** 0.5 * 4)]
bf_case = CoCase(160, 49, 26, 2048, "lm", 11, (2.0 / 2048) ** 0.5 * 4, bf16=True)     # config 4's size, every 4th round
# a busy chip, channel-major features: where round 4's race of the forward kernel's phase 1 showed (one launch in eight at
# this size, one in seventy at B = 160) -- every 4th round
big_cm_case = CoCase(640, 196, 26, 512, "cm", 12, (2.0 / 512) ** 0.5 * 4)
OUT_NAMES = ["v0", "v1", "v2", "q0", "q1", "q2", "dV", "dQ0", "dQ1", "dQ2", "dW_v", "db_v", "dW_q", "db_q", "dw_v", "dc_v",
             "dw_q", "dc_q"]

first = {}
t_start = time.time()
done = 0
for it in range(args.iters):
    if time.time() - t_start > args.budget_s:
        say("time budget reached after %d rounds" % it)
        break
    # stock
    y = sa @ sw
    s = (sa * 1.0001).sum(dim=1)
    report("stock_matmul", it, y, stock_ref, 1e-4)                   # (a library GEMM may differ run to run in the last bits)
    report("stock_rowsum", it, s, None, first=first.setdefault("rowsum", s.clone()))
    # gemm.hip
    for am in (True, False):
        Cm = torch.full((gM, gN), float("nan"), device=dev)
        g = _lib.GemmDesc()
        kw = dict(A=gA[am], B=gW, C=Cm, bias_n=gb, M=gM, N=gN, K=gK, batch=1, b_sk=1, b_sn=gK, c_sm=gN, c_sn=1)
        kw.update(dict(a_sm=1, a_sk=gM) if am else dict(a_sm=gK, a_sk=1))
        for k, v in kw.items():
            setattr(g, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        _lib.check(lib.coattn_gemm_f32(C.byref(g), stream), "gemm")
        report("gemm_f32[a_m=%s]" % am, it, Cm, g_ref[am], 2e-6, first.get(("g", am)))
        first.setdefault(("g", am), Cm.clone())
    # gemm_w.hip
    y = torch.full((lM, lN), float("nan"), device=dev)
    _lib.check(lib.coattn_linear_forward(lx.data_ptr(), lK, lW.data_ptr(), lb.data_ptr(), y.data_ptr(), l_wimg.data_ptr(),
                                         lM, lN, lK, 0.0, 0, stream), "linear")
    report("gemm_w", it, y, l_ref, 2e-6, first.get("l"))
    first.setdefault("l", y.clone())
    # gemm_tn.hip
    dW = torch.full((512, 512), float("nan"), device=dev)
    _lib.check(lib.coattn_linear_weight_grad(tdy.data_ptr(), 512, tx.data_ptr(), 512, dW.data_ptr(), t_ws.data_ptr(), lM, 512,
                                             512, 0, stream), "wgrad")
    report("gemm_tn", it, dW, t_ref, 2e-6, first.get("t"))
    first.setdefault("t", dW.clone())
    # gemm_bf.hip (reduced-precision mode), every 4th round (the float64 compare of 2 x 16M elements is the cost)
    if it % 4 == 0:
        y = torch.full((bM, bd), float("nan"), device=dev)
        _lib.check(lib.coattn_linear_forward(bx.data_ptr(), bd, bW.data_ptr(), bb.data_ptr(), y.data_ptr(), b_wimg.data_ptr(),
                                             bM, bd, bd, 0.0, _lib.FLAG_BF16_PROJ, stream), "linear bf16")
        report("gemm_bf_nt", it, y, b_ref, 2e-5, first.get("bn"))
        first.setdefault("bn", y.clone())
        dW = torch.full((bd, bd), float("nan"), device=dev)
        _lib.check(lib.coattn_linear_weight_grad(bdy.data_ptr(), bd, bx.data_ptr(), bd, dW.data_ptr(), bt_ws.data_ptr(), bM, bd,
                                                 bd, _lib.FLAG_BF16_PROJ, stream), "wgrad bf16")
        report("gemm_bf_tn", it, dW, bt_ref, 2e-5, first.get("bt"))
        first.setdefault("bt", dW.clone())
    # fused co-attention forward + backward
    for c in cases + ([bf_case, big_cm_case] if it % 4 == 0 else []):
        outs = c.run()
        for nm, o, r in zip(OUT_NAMES, outs, c.ref):
            if o is None:                                            # (no dV in the frozen-encoder case)
                continue
            if nm in ("dc_v", "dc_q"):                               # identically ~0 (softmax shift invariance): rounding noise,
                r = None                                             # held to repeatability only
            report(c.name + "." + nm, it, o, r, c.tol, c.first[OUT_NAMES.index(nm)] if c.first else None)
        if c.first is None:
            c.first = [o.clone() if o is not None else None for o in outs]
    torch.cuda.synchronize()
    done = it + 1
    if (it + 1) % 250 == 0:
        say("round %d ok so far: failures %s, %.0f s" % (it + 1, json.dumps(failures), time.time() - t_start))

say("clocks at end: %s" % json.dumps(clocks()))
say("soak_kernels: %d rounds on GPU uuid %s: %s" % (done, uuid, "ZERO MISMATCHES" if not failures else "FAILURES " + json.dumps(failures)))
if args.log:
    path = args.log.replace("<uuid>", uuid.replace("-", "")[:16])
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
sys.exit(1 if failures else 0)
