"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the callers/feeders of the co-attention path:
the whole ``--model attention`` network and one training step, used (a) as the checker of the
product's stock-PyTorch feeder modules and HIP path at network level, (b) as the
``cpu_baseline`` ("port") that ``bench.py`` times on the host cores.

Follows /root/reference: HierarchicalCoAttentionNet model.py:157-187, ImageCoAttentionEncoder
model.py:190-243 (VGG11-bn ``features``; torchvision cfg "A" + BatchNorm), QuestionCoAttention-
Encoder model.py:246-298, PhraseConvPool model.py:301-334 (with its channel-grouping quirk),
MLPClassifier model.py:400-434, train step main.py:178-222.  Parity status: PINNED by
``oracle/make_golden_net.py`` (logits / loss trajectory of the imported reference classes).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .coattn_oracle import OracleMLPClassifier, OracleParallelCoAttention


def _vgg11_bn_features():
    layers, c = [], 3
    for v in (64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"):
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
            c = v
    return nn.Sequential(*layers)


class OracleImageEncoder(nn.Module):
    def __init__(self, is_trainable=False, weights_path=None):
        super().__init__()
        self.vgg11_encoder = _vgg11_bn_features()
        if not is_trainable:
            for p in self.vgg11_encoder.parameters():
                p.requires_grad = False

    def forward(self, x):
        return self.vgg11_encoder(x).flatten(2, 3).permute(0, 2, 1)


class OraclePhraseConvPool(nn.Module):
    def __init__(self, e):
        super().__init__()
        self.conv_unigram = nn.Sequential(nn.ConstantPad1d((0, 0), 0), nn.Conv1d(e, e, 1, 1), nn.Tanh())
        self.conv_bigram = nn.Sequential(nn.ConstantPad1d((1, 0), 0), nn.Conv1d(e, e, 2, 1), nn.Tanh())
        self.conv_trigram = nn.Sequential(nn.ConstantPad1d((1, 1), 0), nn.Conv1d(e, e, 3, 1), nn.Tanh())

    def forward(self, x):
        B, T, E = x.shape
        xt = x.permute(0, 2, 1)
        cat = torch.cat([self.conv_unigram(xt), self.conv_bigram(xt), self.conv_trigram(xt)], 1)   # [B,3E,T]
        # out[..., e] = max(cat[3e], cat[3e+1], cat[3e+2])  (model.py:327-332)
        return torch.nn.functional.max_pool1d(cat.permute(0, 2, 1), kernel_size=3, stride=3)


class OracleQuestionEncoder(nn.Module):
    def __init__(self, vocab_size, word_emb_dim, hidden_dim):
        super().__init__()
        self.word_embedding = nn.Embedding(vocab_size, word_emb_dim, padding_idx=0)
        self.phrase_conv_pool = OraclePhraseConvPool(word_emb_dim)
        self.sentence_lstm = nn.LSTM(word_emb_dim, hidden_dim)

    def forward(self, x, lens):
        T = x.shape[1]
        w = self.word_embedding(x)
        p = pack_padded_sequence(self.phrase_conv_pool(w), lens.cpu(), batch_first=True)
        s, _ = self.sentence_lstm(p)
        p = pad_packed_sequence(p, batch_first=True, total_length=T)[0]
        s = pad_packed_sequence(s, batch_first=True, total_length=T)[0]
        return w, p, s


class OracleHierarchicalCoAttentionNet(nn.Module):
    def __init__(self, ques_enc_params, img_enc_params, K, mlp_dim=1024, as_executed=True):
        super().__init__()
        h = ques_enc_params["hidden_dim"]
        self.image_encoder = OracleImageEncoder(**img_enc_params)
        self.question_encoder = OracleQuestionEncoder(**ques_enc_params)
        self.co_attention = OracleParallelCoAttention(h, as_executed=as_executed)
        self.mlp_classify = OracleMLPClassifier(h, mlp_dim, K)

    def forward(self, x_img, x_ques, x_ques_lens):
        qs = list(self.question_encoder(x_ques, x_ques_lens))
        v = self.image_encoder(x_img)
        va, qa = self.co_attention(v, qs)
        return self.mlp_classify(va, qa)


def train_steps(model, batches, lr=1e-4):
    """main.py:178-222 on CPU: CrossEntropyLoss (mean) + Adam(lr) defaults; returns the losses."""
    crit = nn.CrossEntropyLoss()
    opt = torch.optim.Adam(model.parameters(), lr)
    losses = []
    for image, question, ques_len, label in batches:
        loss = crit(model(image, question, ques_len), label)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses
