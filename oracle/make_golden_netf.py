"""TEST INFRASTRUCTURE ONLY -- network-level golden vectors with INJECTED image features, from the imported
reference (container only):   python -m oracle.make_golden_netf

The reference's ``HierarchicalCoAttentionNet`` (model.py:157-187) is built with a closed-form state_dict and its
``image_encoder`` attribute is replaced by a module that returns given features ("stub VGG features
injected"); everything after it -- question hierarchy (Embedding, PhraseConvPool with its grouping quirk, LSTM),
ParallelCoAttention, MLPClassifier, CrossEntropyLoss + Adam(lr) of main.py:178-222 -- is the reference's own code.
Outputs tests/golden/netf_cases.npz: logits and the loss trajectory for NETF_CASE (hidden 512, 7x7 grid) and
NETF4_CASE (BASELINE config 4: 7x7x2048 features, hidden 2048, 3001 logits), float32 reference values, plus
per-parameter checksums after the steps.  The oracle restatement must reproduce them (asserted)."""
import os
import warnings

import numpy as np
import torch

from . import net_oracle as NO
from .golden_cases import NETF4_CASE, NETF_CASE, closed_form_state, netf_inputs
from .ref_import import import_reference_model

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class Inject(torch.nn.Module):
    """Stands in for the image encoder: returns the features it was given."""

    def __init__(self, feats):
        super().__init__()
        self.feats = feats

    def forward(self, _x):
        return self.feats


def run_case(cls, c, tag, out, f64=False):
    """f64: the same modules and fp32 state / inputs computed in float64 (net.double()): the values the fp32 run rounds --
    what the GPU tests hold the HIP path to at the contract's 1e-4 (keys <tag>_logits64 / _losses64)."""
    qp = dict(vocab_size=c["vocab"], word_emb_dim=c["hidden"], hidden_dim=c["hidden"])
    ip = dict(is_trainable=False, weights_path=None)
    net = cls(qp, ip, K=c["K"] + 1)
    sd = closed_form_state(net, c["seed"])
    net.load_state_dict(sd)
    feats, question, lens, label = netf_inputs(c)
    if f64:
        net = net.double()
        feats = feats.double()
        tag = tag + "@64"
    net.image_encoder = Inject(feats)
    net.train()
    dummy = torch.zeros(c["B"], 3, 8, 8)
    logits = net(dummy, question, lens).detach()
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], c["lr"])
    losses = []
    for _ in range(c["steps"]):                       # main.py:211-222
        loss = crit(net(dummy, question, lens), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    if f64:
        out[tag[:-3] + "_logits64"] = logits.numpy()
        out[tag[:-3] + "_losses64"] = np.asarray(losses, dtype=np.float64)
        return logits, losses
    out[tag + "_logits"] = logits.numpy()
    out[tag + "_losses"] = np.asarray(losses, dtype=np.float64)
    after = {k: v for k, v in net.state_dict().items() if "image_encoder" not in k}
    for k, t in after.items():
        if t.dtype.is_floating_point:
            out[tag + "_dabs." + k] = np.float64((t.double() - sd[k].double()).abs().sum())
    return logits, losses


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    ref = import_reference_model()
    out = {}
    for c, tag in ((NETF_CASE, "f2"), (NETF4_CASE, "f4")):
        logits, losses = run_case(ref.HierarchicalCoAttentionNet, c, tag, out)
        chk = {}
        ol, olosses = run_case(NO.OracleHierarchicalCoAttentionNet, c, tag, chk)     # the restatement, same protocol
        e_l = (ol - logits).abs().max().item()
        e_t = max(abs(a - b) for a, b in zip(losses, olosses))
        print(tag, "logits", tuple(logits.shape), "max|logit| %.3f" % logits.abs().max().item(), "losses", losses,
              "| oracle err logits %.2e losses %.2e" % (e_l, e_t))
        assert e_l < 2e-5 and e_t < 2e-5, (tag, e_l, e_t)
        l64, t64 = run_case(ref.HierarchicalCoAttentionNet, c, tag, out, f64=True)   # the reference itself in float64
        print(tag, "float64 reference: fp32 run off by logits %.2e losses %.2e"
              % ((l64 - logits.double()).abs().max().item(), max(abs(a - b) for a, b in zip(losses, t64))))
    np.savez_compressed(os.path.join(OUT_DIR, "netf_cases.npz"), **out)


if __name__ == "__main__":
    main()
