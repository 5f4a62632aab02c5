import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _golden as G
from tests._hip import run_hip
V, Qs, P, gv, gq = G.build_case("g3_n49_ragged", torch.float32)
a = run_hip(V, Qs, P, gv, gq, impl="fused", layout="cm")
b = run_hip(V, Qs, P, gv, gq, impl="general", layout="cm")
for k in ("dV_phys", "dQ", "dW_v.weight", "dW_v.bias", "dW_q.weight"):
    e = (a[k] - b[k]).abs()
    print(k, "max err %.3e of %.3e at" % (e.max().item(), b[k].abs().max().item()), (e == e.max()).nonzero()[0].tolist(),
          "frac bad %.4f" % (e > 1e-4 * b[k].abs().max()).float().mean().item())
e = (a["dW_v.weight"] - b["dW_v.weight"]).abs()
bad = (e > 1e-4 * b["dW_v.weight"].abs().max())
print("bad rows (j):", bad.any(1).nonzero().flatten().tolist()[:40])
print("bad cols (k):", bad.any(0).nonzero().flatten().tolist()[:40])
e = (a["dV_phys"] - b["dV_phys"]).abs()
bad = (e > 1e-4 * b["dV_phys"].abs().max())
print("dV bad b:", bad.any(2).any(1).nonzero().flatten().tolist(), "bad k count", bad.any(2).any(0).sum().item(), "bad n:", bad.any(1).any(0).nonzero().flatten().tolist())
