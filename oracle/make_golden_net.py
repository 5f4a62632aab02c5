"""TEST INFRASTRUCTURE ONLY -- network-level golden vectors from the imported reference
(container only):   python -m oracle.make_golden_net

G7: logits of the reference's ``HierarchicalCoAttentionNet`` (model.py:157-187; random-init
    VGG11-bn features from the torchvision stub, closed-form state_dict) on a tiny batch --
    pins the wiring, the PhraseConvPool channel-grouping quirk, K+1 logits and the state_dict
    key list.
G8: train-step golden for main.py:178-222: CrossEntropyLoss + Adam(lr=1e-4), 3 steps on CPU
    with the reference classes -> losses and per-parameter checksums; W_b bit-identical to init.
G9: logits of the reference's ``VQABaselineNet`` question branch + MLP with injected 4096-d
    image vectors in eval() mode (BASELINE config 1, K=2 -> 3 logits).
"""
import json
import os
import warnings

import numpy as np
import torch

from . import coattn_oracle as O
from . import net_oracle as NO
from .golden_cases import BASE_CASE, NET_CASE, closed_form_state, net_case_batch
from .ref_import import import_reference_model

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(8)
    ref = import_reference_model()
    c = NET_CASE
    qp = dict(vocab_size=c["vocab"], word_emb_dim=c["hidden"], hidden_dim=c["hidden"])
    ip = dict(is_trainable=False, weights_path=None)
    net = ref.HierarchicalCoAttentionNet(qp, ip, K=c["K"] + 1)
    sd = closed_form_state(net, c["seed"])
    net.load_state_dict(sd)
    keys = list(net.state_dict().keys())
    batch = net_case_batch()
    net.train()
    out = {}
    with torch.no_grad():
        feats = net.image_encoder(batch[0])
        out["img_feat_strides"] = np.array(feats.stride())
        out["img_feat_shape"] = np.array(feats.shape)
    # the forward above updated BatchNorm running stats: reload so that G7/G8 start from sd
    net.load_state_dict(sd)
    logits = net(batch[0], batch[1], batch[2])
    out["g7_logits"] = logits.detach().numpy()
    net.load_state_dict(sd)
    losses = NO.train_steps(net, [batch] * c["steps"], lr=c["lr"])     # generic loop of main.py:178-222
    out["g8_losses"] = np.array(losses)
    after = net.state_dict()
    out["g8_wb_unchanged"] = np.array(bool(torch.equal(after["co_attention.W_b.weight"], sd["co_attention.W_b.weight"])))
    for k in keys:
        t = after[k]
        if t.dtype.is_floating_point and "vgg11_encoder" not in k:
            out["g8_sum." + k] = np.float64(t.double().sum())
            out["g8_dabs." + k] = np.float64((t.double() - sd[k].double()).abs().sum())
    # oracle restatement must reproduce both
    onet = NO.OracleHierarchicalCoAttentionNet(qp, ip, K=c["K"] + 1)
    onet.load_state_dict(sd)
    ol = onet(batch[0], batch[1], batch[2])
    e7 = (ol - logits).abs().max().item()
    onet.load_state_dict(sd)
    olosses = NO.train_steps(onet, [batch] * c["steps"], lr=c["lr"])
    e8 = max(abs(a - b) for a, b in zip(losses, olosses))
    assert e7 < 1e-5 and e8 < 1e-5, (e7, e8)
    # G9 baseline question branch + MLP with injected image embedding
    b = BASE_CASE
    bnet_q = ref.QuestionBaselineEncoder(b["vocab"], b["emb"], b["hidden"])
    mlp = torch.nn.Sequential(torch.nn.Linear(1024, 1000), torch.nn.Dropout(0.5), torch.nn.Tanh())
    fc = torch.nn.Linear(1000, b["K"] + 1)
    holder = torch.nn.ModuleDict({"question_encoder": bnet_q, "mlp": mlp, "fc_final": fc})
    holder.load_state_dict(closed_form_state(holder, b["seed"]))
    holder.eval()
    tok = (O.hash_uniform(b["B"] * b["T"], b["seed"] + 2) * (b["vocab"] - 1)).astype("int64").reshape(b["B"], b["T"]) + 1
    lens = torch.tensor(b["lens"])
    img_emb = torch.tanh(torch.from_numpy(O.hash_normal((b["B"], 1024), b["seed"] + 5)).float())
    with torch.no_grad():
        out["g9_logits"] = fc(mlp(img_emb * bnet_q(torch.from_numpy(tok), lens))).numpy()
    np.savez_compressed(os.path.join(OUT_DIR, "net_cases.npz"), **out)
    with open(os.path.join(OUT_DIR, "net_state_keys.json"), "w") as fh:
        json.dump({"attention_state_dict_keys": keys, "oracle_vs_ref_logits": e7, "oracle_vs_ref_losses": e8,
                   "losses": losses}, fh, indent=1)
    print("G7 logits", logits.shape, "oracle err", e7, "| G8 losses", losses, "oracle err", e8)
    print("feature strides", feats.stride(), feats.shape)


if __name__ == "__main__":
    main()
