"""TEST INFRASTRUCTURE ONLY -- the table of golden cases for the co-attention path and the
closed-form regeneration of their inputs (so that ``tests/golden/*.npz`` hold outputs only).

Case ids follow SURVEY.md section 8c (G1..G6); every case pins forward outputs, level-0
intermediates and every gradient of ``ParallelCoAttention`` (model.py:337-397) produced by
the imported reference in float32 (the reference's CPU path) and in float64.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from . import coattn_oracle as O

# name -> dict(B,N,T,d, lens, scale_v, scale_q, seed, full)   (full: store whole tensors)
CASES: Dict[str, dict] = {
    # G1 tiny, ragged lengths incl. len=1 (96.5 % of a_q's mass lands on pad rows)
    "g1_tiny_d64": dict(B=3, N=49, T=26, d=64, lens=[26, 9, 1], scale_v=0.5, scale_q=0.5, seed=101, full=True),
    # odd shapes for the general-shape kernels (N,T,d not multiples of the tile sizes)
    "g1_odd_d96": dict(B=2, N=37, T=5, d=96, lens=[5, 2], scale_v=0.4, scale_q=0.4, seed=102, full=True),
    # G2 headline kernel shape (448x448 -> 14x14 grid), natural scale: tanh saturates
    "g2_cfg2_natural": dict(B=2, N=196, T=26, d=512, lens=[26, 7], scale_v=1.0, scale_q=1.0, seed=103, full=False),
    # G5 same shape, 1/sqrt(d)-scaled: unsaturated tanh, exercises the dA path
    "g5_cfg2_scaled": dict(B=2, N=196, T=26, d=512, lens=[26, 7], scale_v=1.0,
                           scale_q=math.sqrt(2.0 / 512), seed=104, full=False),
    # G3/G4 224x224 -> 7x7 grid, ragged incl. all-26 and len=1
    "g3_n49_ragged": dict(B=3, N=49, T=26, d=512, lens=[26, 26, 1], scale_v=1.0,
                          scale_q=math.sqrt(2.0 / 512), seed=105, full=False),
    # d=256 (two-wave configuration of the fused kernels)
    "g5_d256": dict(B=2, N=196, T=26, d=256, lens=[20, 3], scale_v=1.0,
                    scale_q=math.sqrt(2.0 / 256), seed=106, full=False),
    # G6 ResNet-like 7x7x2048 features (fp32 oracle for BASELINE config 4)
    "g6_d2048_n49": dict(B=2, N=49, T=26, d=2048, lens=[26, 11], scale_v=1.0,
                         scale_q=math.sqrt(2.0 / 2048), seed=107, full=False),
}


def build_case(name: str, dtype=torch.float32):
    """Regenerates (V_phys, Qs, params, gv, gq) of a case in closed form."""
    c = CASES[name]
    P = O.make_params(c["d"], c["seed"], dtype)
    V, Qs = O.make_inputs(c["B"], c["N"], c["T"], c["d"], c["seed"], lens=c["lens"],
                          scale_v=c["scale_v"], scale_q=c["scale_q"], dtype=dtype)
    gv = torch.from_numpy(O.hash_normal((3, c["B"], c["d"]), c["seed"] + 900)).to(dtype)
    gq = torch.from_numpy(O.hash_normal((3, c["B"], c["d"]), c["seed"] + 901)).to(dtype)
    return V, Qs, P, gv, gq


GRAD_KEYS = ("dV_phys", "dQ") + tuple("d" + k for k in O.PARAM_KEYS)
