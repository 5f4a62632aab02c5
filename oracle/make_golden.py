"""TEST INFRASTRUCTURE ONLY -- generate ``tests/golden/*.npz`` from the imported reference.

Run in the build container only (needs ``/root/reference``):

    python -m oracle.make_golden            # writes tests/golden/<case>.npz + MANIFEST.json

For every case of ``oracle/golden_cases.py`` the reference class ``ParallelCoAttention``
(model.py:337-397) is instantiated, loaded with the closed-form state_dict, and run on CPU in
float32 (= the reference's CPU path, the parity target) and in float64 (to separate our
rounding from the reference's).  Intermediates are captured by wrapping the ``F.tanh`` /
``F.softmax`` the reference calls (model.py:377-388), so they are the reference's own values.
While generating, the oracle restatement (forward + hand-derived backward) is checked against
the same run; the script aborts if that check fails, i.e. a written fixture also certifies
the oracle.
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import torch

from . import coattn_oracle as O
from .golden_cases import CASES, GRAD_KEYS, build_case
from .ref_import import import_reference_model

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def run_reference(ref, name, dtype):
    c = CASES[name]
    V, Qs, P, gv, gq = build_case(name, dtype)
    m = ref.ParallelCoAttention(c["d"]).to(dtype)
    m.load_state_dict(P)
    V = V.clone().requires_grad_(True)
    Qs = [q.clone().requires_grad_(True) for q in Qs]
    cap = {"tanh": [], "softmax": []}
    o_tanh, o_soft = ref.F.tanh, ref.F.softmax

    def tanh_rec(x):
        y = o_tanh(x); cap["tanh"].append(y.detach()); return y

    def soft_rec(x, dim):
        y = o_soft(x, dim=dim); cap["softmax"].append(y.detach()); return y

    ref.F.tanh, ref.F.softmax = tanh_rec, soft_rec
    try:
        vs, qs = m(V.permute(0, 2, 1), Qs)      # x_img is the permuted view, model.py:217
    finally:
        ref.F.tanh, ref.F.softmax = o_tanh, o_soft
    loss = sum((vs[l] * gv[l]).sum() + (qs[l] * gq[l]).sum() for l in range(3))
    loss.backward()
    assert m.W_b.weight.grad is None and m.W_b.bias.grad is None      # dead layer, SURVEY 0.1
    res = {"v": torch.stack(vs).detach(), "q": torch.stack(qs).detach(), "loss": loss.detach(),
           "C": torch.stack(cap["tanh"][0::3]),                        # per level: C, H_v, H_q
           "H_q": torch.stack(cap["tanh"][2::3]),
           "a_v": torch.stack(cap["softmax"][0::2]).squeeze(3),
           "a_q": torch.stack(cap["softmax"][1::2]).squeeze(3),
           "dV_phys": V.grad.detach(), "dQ": torch.stack([q.grad for q in Qs]).detach()}
    for k, p in m.named_parameters():
        if not k.startswith("W_b"):
            res["d" + k] = p.grad.detach()
    return res


def check_oracle(name, dtype, res, tol):
    V, Qs, P, gv, gq = build_case(name, dtype)
    f = O.coattn_forward(V, Qs, P)
    g = O.coattn_backward(V, Qs, P, gv, gq)
    worst = 0.0
    for k in ("v", "q", "C", "a_v", "a_q", "H_q"):
        worst = max(worst, (f[k] - res[k]).abs().max().item())
    for k in GRAD_KEYS:
        scale = max(1.0, res[k].abs().max().item())
        worst = max(worst, (g[k] - res[k]).abs().max().item() / scale)
    if not worst <= tol:
        raise SystemExit("oracle restatement disagrees with the reference on %s (%s): %g" % (name, dtype, worst))
    return worst


def main():
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference_model()
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {"generator": "python -m oracle.make_golden", "torch": torch.__version__,
                "reference": "model.py ParallelCoAttention (model.py:337-397), CPU", "cases": {}}
    for name, c in CASES.items():
        r32 = run_reference(ref, name, torch.float32)
        r64 = run_reference(ref, name, torch.float64)
        e32 = check_oracle(name, torch.float32, r32, 2e-5)
        e64 = check_oracle(name, torch.float64, r64, 1e-12)
        out = {}
        for tag, r in (("32", r32), ("64", r64)):
            for k in ("v", "q", "loss", "a_v", "a_q"):
                out[k + tag] = r[k].numpy()
            out["C0_" + tag] = r["C"][0].numpy().astype(np.float32)
            out["Hq0_" + tag] = r["H_q"][0].numpy().astype(np.float32)
            for k in GRAD_KEYS:
                if c["full"] or r[k].numel() <= 4096:
                    out["g%s.%s" % (tag, k)] = r[k].numpy().astype(np.float32 if tag == "32" else np.float64)
                ck = O.checksum(r[k])
                for f, val in ck.items():
                    out["ck%s.%s.%s" % (tag, k, f)] = val
        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **out)
        diff = max((r32[k].double() - r64[k]).abs().max().item() for k in ("v", "q"))
        manifest["cases"][name] = dict(c, oracle_err_f32=e32, oracle_err_f64=e64,
                                       ref_f32_vs_f64_fwd=diff, bytes=os.path.getsize(path))
        print("%-18s oracle-vs-ref f32 %.2e f64 %.2e | ref f32-vs-f64 fwd %.2e | %d B"
              % (name, e32, e64, diff, os.path.getsize(path)))
    with open(os.path.join(OUT_DIR, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
