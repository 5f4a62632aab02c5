"""GPU: the HIP path INSIDE the network at the op-level tolerance.  Image features are injected (closed form), so
nothing stock and batch-statistics dependent (MIOpen convolutions, 3-sample BatchNorm) sits in front of the path:
question hierarchy (stock Embedding / LSTM + HIP PhraseConvPool) -> HIP co-attention -> MLPClassifier -> HIP cross
entropy -> Adam, against goldens of the imported reference run the same way (oracle/make_golden_netf.py;
SURVEY.md 8c G7/G8 "stub VGG features injected"), model.py:171-187, main.py:211-222.

  f2: hidden 512, 7x7 grid (cfg-2 like): logits and 3-step loss trajectory <= 1e-4, both feature layouts.
  f4: BASELINE config 4 (7x7x2048 features, hidden 2048, 3001 logits): fp32 mode <= 1e-4 on logits / losses against the
      reference run in FLOAT64 (keys *_logits64 / *_losses64), in the tolerance mode train.Trainer runs AND in the exact mode;
      bf16 mode (autocast around the stock encoders + bf16-MFMA projections) within bf16 tolerance of the fp32
      reference (stated below).
  cfg 5 (frozen BERT-768 token embeddings as word level, an extension without a reference): HIP path vs the same
      modules on the CPU with the oracle's co-attention / MLP, logits and losses <= 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle.golden_cases import NETF4_CASE, NETF_CASE, closed_form_state, netf_inputs
from tests._golden import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _net(c, fast=True):
    """fast: the tolerance mode of the fp32 products (what train.Trainer sets; include/coattn.h COATTN_FLAG_FAST16) on the
    co-attention and the phrase level; False: the modules' default, fp32-accurate products."""
    from vqa_amd.modules import HierarchicalCoAttentionNet
    qp = dict(vocab_size=c["vocab"], word_emb_dim=c["hidden"], hidden_dim=c["hidden"])
    net = HierarchicalCoAttentionNet(qp, dict(is_trainable=False, weights_path=None), K=c["K"] + 1)
    sd = closed_form_state(net, c["seed"])
    net.load_state_dict(sd)
    net.co_attention.fast_products = fast
    net.question_encoder.phrase_conv_pool.fast_products = fast
    return net.cuda(), sd


def _steps(net, feats, question, lens, label, c, autocast=False):
    """main.py:211-222 from the features on: logits, then `steps` Adam steps; returns (logits, losses)."""
    from vqa_amd.loss import CrossEntropyLoss
    import contextlib
    ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    with torch.no_grad(), ctx():
        logits = net.forward_features(feats, question, lens).float().cpu().numpy()
    crit = CrossEntropyLoss()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], c["lr"])
    losses = []
    for _ in range(c["steps"]):
        with ctx():
            out = net.forward_features(feats, question, lens)
        loss = crit(out.float(), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return logits, np.asarray(losses)


@pytest.mark.parametrize("layout", ["cm", "lm"])
def test_f2_injected_features_logits_and_trajectory(layout):
    c = NETF_CASE
    gold = np.load(os.path.join(GOLDEN_DIR, "netf_cases.npz"))
    net, sd = _net(c)
    feats, question, lens, label = netf_inputs(c)
    feats = feats.cuda()                                        # permuted view of [B,d,N] (the reference's layout)
    if layout == "lm":
        feats = feats.contiguous()                              # what a channels_last encoder hands over
    logits, losses = _steps(net, feats, question.cuda(), lens, label.cuda(), c)
    e_l = np.abs(logits - gold["f2_logits64"]).max()
    e_t = np.abs(losses - gold["f2_losses64"]).max()
    print("f2", layout, "logits err %.2e, losses err %.2e (vs the float64 reference run)" % (e_l, e_t), losses)
    assert e_l < 5e-5 and e_t < 5e-5                            # (contract 1e-4; tighter, so that green certifies margin)
    import vqa_amd
    vqa_amd.check_range()                                       # tolerance mode: no operand left the FP16-piece range
    # every trainable parameter moved as in the reference run (sum |delta| after the steps), W_b never
    after = net.state_dict()
    for k in gold.files:
        if k.startswith("f2_dabs."):
            name = k[len("f2_dabs."):]
            got = float((after[name].double().cpu() - sd[name].double()).abs().sum())
            ref = float(gold[k])
            if name.endswith("w_v.bias") or name.endswith("w_q.bias"):
                continue             # analytically zero gradient: Adam turns rounding noise into +-lr steps
            assert abs(got - ref) <= 2e-2 * max(ref, 1e-9) + 1e-7, (name, got, ref)
    assert torch.equal(after["co_attention.W_b.weight"].cpu(), sd["co_attention.W_b.weight"])


def test_f4_config4_fp32_and_bf16():
    c = NETF4_CASE
    gold = np.load(os.path.join(GOLDEN_DIR, "netf_cases.npz"))
    feats, question, lens, label = netf_inputs(c)
    feats = feats.cuda().contiguous()                           # ResNet-like 7x7x2048 grid, location-major
    for fast in (True, False):
        net, _ = _net(c, fast=fast)
        logits, losses = _steps(net, feats, question.cuda(), lens, label.cuda(), c)
        e_l = np.abs(logits - gold["f4_logits64"]).max()
        e_t = np.abs(losses - gold["f4_losses64"]).max()
        print("f4 fp32 (%s mode): logits err %.2e (max |logit| %.2f), losses err %.2e, vs the float64 reference run"
              % ("tolerance" if fast else "exact", e_l, np.abs(gold["f4_logits64"]).max(), e_t))
        assert e_l < 1e-4 and e_t < 1e-4, (fast, e_l, e_t)     # the contract at config 4's own width (was 2e-4 vs an fp32 run)
    # bf16 mode of config 4 (apex O1 analogue): bf16 autocast around the stock modules, co-attention projections on
    # the bf16 MFMA.  Tolerance: bf16 has 8 significant bits; logits are O(1) sums over 1024 bf16 products.
    net, _ = _net(c)
    net.co_attention.bf16_projections = True
    logits, losses = _steps(net, feats, question.cuda(), lens, label.cuda(), c, autocast=True)
    e_l = np.abs(logits - gold["f4_logits"]).max()
    e_t = np.abs(losses - gold["f4_losses"]).max() / np.abs(gold["f4_losses"]).max()
    print("f4 bf16: logits err %.2e, relative losses err %.2e" % (e_l, e_t))
    assert e_l < 5e-2 and e_t < 5e-2


def test_config5_bert_word_level_vs_cpu_oracle():
    """BASELINE config 5: 768-d token embeddings (frozen, random init) -> Linear(768,512) word level.  No reference
    exists (README TODO), so the checker is the same question encoder on the CPU + the oracle's co-attention / MLP."""
    import copy
    from oracle.coattn_oracle import OracleMLPClassifier, OracleParallelCoAttention
    from vqa_amd import train as T
    from vqa_amd.modules import HierarchicalCoAttentionNet
    torch.manual_seed(0)
    qp = T.bert_question_params(hidden_dim=512, vocab_size=120)
    net = HierarchicalCoAttentionNet(qp, dict(is_trainable=False, weights_path=None), K=11)
    c = dict(NETF_CASE, vocab=120)
    feats, question, lens, label = netf_inputs(c)
    # CPU checker: same question encoder (stock torch ops; its PhraseConvPool takes the stock path on CPU tensors)
    q_cpu = copy.deepcopy(net.question_encoder)
    co = OracleParallelCoAttention(512, as_executed=True)
    co.load_state_dict(net.co_attention.state_dict())
    mlp = OracleMLPClassifier(512, net.mlp_classify.W_s.out_features, 11)
    mlp.load_state_dict(net.mlp_classify.state_dict())
    params = [p for m in (q_cpu, co, mlp) for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, 1e-4)
    ref_losses, ref_logits = [], None
    for _ in range(3):
        out = mlp(*co(feats, list(q_cpu(question, lens))))
        if ref_logits is None:
            ref_logits = out.detach().numpy()
        loss = torch.nn.functional.cross_entropy(out, label)
        opt.zero_grad(); loss.backward(); opt.step()
        ref_losses.append(float(loss))
    net = net.cuda()
    logits, losses = _steps(net, feats.cuda(), question.cuda(), lens, label.cuda(), dict(c, lr=1e-4, steps=3))
    e_l = np.abs(logits - ref_logits).max()
    e_t = np.abs(losses - np.asarray(ref_losses)).max()
    print("cfg5: logits err %.2e, losses err %.2e" % (e_l, e_t))
    assert e_l < 1e-4 and e_t < 1e-4
    assert net.question_encoder.word_proj.weight.grad is not None
