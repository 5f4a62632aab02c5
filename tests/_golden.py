"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests: load a golden case and
compare a result dict against it, tensor-by-tensor where the fixture holds the full tensor
and through the stored checksums (sum, sum|x|, linear functionals, sampled entries) otherwise."""
import json
import os

import numpy as np
import torch

from oracle import coattn_oracle as O
from oracle.golden_cases import CASES, GRAD_KEYS, build_case  # noqa: F401

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def manifest():
    with open(os.path.join(GOLDEN_DIR, "MANIFEST.json")) as fh:
        return json.load(fh)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def fwd_errors(res, gold, tag="64"):
    """max |res - reference| for v, q, a_v, a_q and level-0 C / H_q (res: dict of tensors)."""
    out = {}
    for k in ("v", "q", "a_v", "a_q"):
        if k in res:
            out[k] = float(np.abs(res[k].detach().cpu().double().numpy() - gold[k + tag]).max())
    if "C" in res:
        out["C0"] = float(np.abs(res["C"][0].detach().cpu().double().numpy() - gold["C0_" + tag]).max())
    if "H_q" in res:
        out["Hq0"] = float(np.abs(res["H_q"][0].detach().cpu().double().numpy() - gold["Hq0_" + tag]).max())
    return out


def grad_errors(res, gold, tag="64"):
    """Relative error per gradient: max|err| / max(1e-6, max|ref|) on full tensors, and on the
    checksum functionals |proj err| / sum|ref| plus sampled entries otherwise."""
    out = {}
    for k in GRAD_KEYS:
        if k not in res:
            continue
        x = res[k].detach().cpu().double().reshape(-1).numpy()
        full = "g%s.%s" % (tag, k)
        if full in gold.files:
            ref = gold[full].astype(np.float64).reshape(-1)
            # dc_v / dc_q are analytically 0 (softmax shift invariance): absolute error there
            floor = 1.0 if k in ("dw_v.bias", "dw_q.bias") else 1e-6
            out[k] = float(np.abs(x - ref).max() / max(floor, np.abs(ref).max()))
            continue
        pre = "ck%s.%s." % (tag, k)
        assert int(gold[pre + "n"]) == x.size, (k, x.size)
        ck = O.checksum(torch.from_numpy(x))
        samp = gold[pre + "samp"]
        scale = max(1e-6, float(np.abs(samp).max()))
        e_s = float(np.abs(ck["samp"] - samp).max() / scale)
        # functionals: error relative to sum|ref| (the natural scale of a random +-1 projection)
        e_p = float(np.abs(ck["proj"] - gold[pre + "proj"]).max() / max(1e-6, float(gold[pre + "abs"])) * np.sqrt(x.size))
        e_a = float(abs(ck["abs"] - float(gold[pre + "abs"])) / max(1e-6, float(gold[pre + "abs"])))
        out[k] = max(e_s, e_p, e_a)
    return out
