"""Developer probe: dQ / dW_q of the fused backward with the bf16-split dP_q kernel against the exact-f32 one
(COATTN_BWD_DPQ_F32=1), each in its own process."""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from tests._hip import run_hip
    from oracle import coattn_oracle as O
    torch.manual_seed(1)
    B, N, T, d = int(os.environ.get("B", 4)), int(os.environ.get("N", 196)), 26, 512
    V = torch.randn(B, d, N) * 0.3
    Qs = [torch.randn(B, T, d) * 0.3 for _ in range(3)]
    P = O.make_params(d, 3)
    gv, gq = torch.randn(3, B, d), torch.randn(3, B, d)
    if os.environ.get("GQ0"): gq = gq * 0
    if os.environ.get("GV0"): gv = gv * 0
    out = run_hip(V, Qs, P, gv, gq, impl="fused", layout="lm", return_ws=True)
    L = 3
    f64 = lambda n: (n + 63) // 64 * 64
    o_dzq = f64(L * B * N); o_dpq = o_dzq + f64(L * B * T * d)
    ws = out["_ws_bwd"]
    sv = {k: v.cpu() if torch.is_tensor(v) else [t.cpu() for t in v] for k, v in out.items() if k.startswith("d")}
    sv["dZq_ws"] = ws[o_dzq:o_dzq + L * B * T * d].view(L, B, T, d).cpu()
    sv["dPq_ws"] = ws[o_dpq:o_dpq + L * B * T * d].view(L, B, T, d).cpu()
    torch.save(sv, sys.argv[1])
    sys.exit(0)
for tag, env in (("new", {}), ("old", {"COATTN_BWD_DPQ_F32": "1"})):
    subprocess.check_call([sys.executable, __file__, "/tmp/dpq_%s.pt" % tag], env=dict(os.environ, **env))
a, b = torch.load("/tmp/dpq_new.pt"), torch.load("/tmp/dpq_old.pt")
d_new = a["dPq_ws"] - a["dZq_ws"]; d_old = b["dPq_ws"] - b["dZq_ws"]       # the C dZ_v part
print("C dZ_v part, sample 0 level 0, channel 5: new", d_new[0, 0, :, 5].tolist())
print("C dZ_v part, sample 0 level 0, channel 5: old", d_old[0, 0, :, 5].tolist())
print("dPq new ch5:", [round(x, 5) for x in a["dPq_ws"][0, 0, :, 5].tolist()])
print("dZq     ch5:", [round(x, 5) for x in a["dZq_ws"][0, 0, :, 5].tolist()])
print("dPq new ch5 sample1 level2:", [round(x, 5) for x in a["dPq_ws"][2, 1, :, 5].tolist()])
print("dPq new ch37:", [round(x, 5) for x in a["dPq_ws"][0, 0, :, 37].tolist()])
print("dZq     ch37:", [round(x, 5) for x in a["dZq_ws"][0, 0, :, 37].tolist()])
for k in a:
    xs = a[k] if isinstance(a[k], list) else [a[k]]
    ys = b[k] if isinstance(b[k], list) else [b[k]]
    for i, (x, y) in enumerate(zip(xs, ys)):
        e = (x - y).abs()
        print(k, i, "max err %.3e  max ref %.3e" % (e.max().item(), y.abs().max().item()))
        if e.max() > 1e-3 * y.abs().max() and x.dim() >= 3:
            bad = (e > 1e-3 * y.abs().max())
            idx = bad.nonzero()
            print("   shape", tuple(x.shape), "bad count", int(bad.sum()), "of", bad.numel(), "first", idx[:4].tolist())
            print("   t values", sorted(set(idx[:, -2].tolist()))[:40], " channel values", sorted(set(idx[:, -1].tolist()))[:24], "...")
