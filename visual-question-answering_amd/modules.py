"""Callers and feeders of the co-attention path, on stock PyTorch-ROCm.

Everything here is ordinary ``torch.nn`` (MIOpen convolutions / LSTM, hipBLASLt GEMMs): the
image and question encoders stay stock by design (BASELINE.json north_star); only
``co_attention`` is the hand-written HIP path.  Class names, constructor arguments, submodule
attribute names (= ``state_dict`` keys) and ``forward`` signatures follow the reference so that
its checkpoints load and its training loop drives these modules unchanged:

  HierarchicalCoAttentionNet   model.py:157-187      VQABaselineNet            model.py:10-38
  ImageCoAttentionEncoder      model.py:190-243      ImageBaselineEncoder      model.py:41-105
  QuestionCoAttentionEncoder   model.py:246-298      QuestionBaselineEncoder   model.py:108-151
  PhraseConvPool               model.py:301-334      MLPClassifier             model.py:400-434

(PhraseConvPool and MLPClassifier keep the reference's submodules and keys but compute through the HIP library on
CUDA tensors: phrase.py, head.py.)

torchvision is not installed here, so the VGG11-bn topology (cfg "A" + BatchNorm) is spelled
out; a torchvision ``vgg11_bn`` state_dict file loads into it through ``weights_path``.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .coattention import ParallelCoAttention
from .head import answer_head

_VGG11_CFG = (64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M")


def vgg11_bn_features() -> nn.Sequential:
    """conv3x3-BN-ReLU stacks with five 2x2 max-pools: 448x448 -> 14x14x512, 224x224 -> 7x7x512."""
    mods, c_in = [], 3
    for item in _VGG11_CFG:
        if item == "M":
            mods.append(nn.MaxPool2d(2, 2))
            continue
        mods.extend((nn.Conv2d(c_in, item, 3, padding=1), nn.BatchNorm2d(item), nn.ReLU(inplace=True)))
        c_in = item
    return nn.Sequential(*mods)


def _vgg11_bn_classifier_head() -> list:
    """torchvision's classifier minus the final FC-1000 (model.py:93)."""
    return [nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(True),
            nn.Dropout()]


def _load_vgg_weights(features: nn.Module, classifier: nn.Module | None, weights_path):
    """Load a torchvision vgg11_bn state_dict file if one is given and present (model.py:232-233).
    Without a file the reference would download pretrained weights; there is no network here, so
    the encoder keeps its random init."""
    if not weights_path or not os.path.isfile(str(weights_path)):
        return False
    sd = torch.load(weights_path, map_location="cpu")
    features.load_state_dict({k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")})
    if classifier is not None:
        cls = {k[len("classifier."):]: v for k, v in sd.items() if k.startswith("classifier.")}
        classifier.load_state_dict({k: v for k, v in cls.items() if not k.startswith("6.")}, strict=False)
    return True


def run_conv_bn_stack(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Runs a Conv2d / BatchNorm2d / ReLU / MaxPool2d ``Sequential`` (the VGG ``features``) with the
    same stock kernels but two value-preserving graph rewrites that save whole elementwise passes
    over the largest activations:

    * ``MaxPool2d(ReLU(x)) == ReLU(MaxPool2d(x))`` exactly (both are monotonic): the ReLU runs on the
      4x smaller pooled tensor.
    * A frozen Conv2d bias in front of a BatchNorm2d that normalises with BATCH statistics (train
      mode -- the reference never calls ``.eval()`` on its frozen VGG, model.py:239-241) cancels in
      ``(y + b) - mean(y + b)``: the convolution is issued without bias (saves the separate
      broadcast-add kernel) and ``momentum * b`` is added to ``running_mean`` instead, which is the
      only place the bias survives (running_var is shift-invariant).  Outputs differ from the plain
      graph by fp32 rounding only (tests/test_net_cpu.py).

    Anything that does not match these patterns (eval-mode BatchNorm, trainable bias, other modules)
    is executed as is.  ``VQA_ENCODER_REWRITE=0`` disables both rewrites."""
    mods = list(seq)
    if os.environ.get("VQA_ENCODER_REWRITE", "1") == "0":
        return seq(x)
    i, n = 0, len(mods)
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if (isinstance(m, nn.Conv2d) and m.bias is not None and not m.bias.requires_grad and m.padding_mode == "zeros"
                and isinstance(nxt, nn.BatchNorm2d) and nxt.training and nxt.momentum is not None):
            x = nxt(F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups))
            if nxt.track_running_stats and nxt.running_mean is not None:
                with torch.no_grad():
                    nxt.running_mean.add_(m.bias.to(nxt.running_mean.dtype), alpha=nxt.momentum)
            i += 2
        elif isinstance(m, nn.ReLU) and isinstance(nxt, nn.MaxPool2d):
            x = F.relu(nxt(x), inplace=True)
            i += 2
        else:
            x = m(x)
            i += 1
    return x


class ImageCoAttentionEncoder(nn.Module):
    """VGG11-bn ``features`` -> spatial grid [B, N, 512] (a permuted view of [B, 512, N])."""

    def __init__(self, is_trainable, weights_path):
        super().__init__()
        self.is_trainable = is_trainable
        self.weights_path = weights_path
        self.vgg11_encoder = vgg11_bn_features()
        _load_vgg_weights(self.vgg11_encoder, None, weights_path)
        self.flatten = nn.Flatten(start_dim=2, end_dim=3)
        if not is_trainable:                      # frozen; BatchNorm stays in train mode (model.py:239-241)
            for prm in self.vgg11_encoder.parameters():
                prm.requires_grad = False

    def forward(self, x_img):
        grid = self.flatten(run_conv_bn_stack(self.vgg11_encoder, x_img))      # [B, 512, N]
        return grid.permute(0, 2, 1)                         # [B, N, 512] view, strides (512N, 1, N)


class _Bottleneck(nn.Module):
    """torchvision's ResNet bottleneck (1x1 -> 3x3 (stride) -> 1x1 x4, BatchNorm, residual)."""

    def __init__(self, c_in, width, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(c_in, width, 1, bias=False); self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, 4 * width, 1, bias=False); self.bn3 = nn.BatchNorm2d(4 * width)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


def resnet152_features() -> nn.Sequential:
    """ResNet-152 trunk (stem + layers [3, 8, 36, 3]) up to the 2048-channel stride-32 map:
    224x224 -> 7x7x2048 (BASELINE config 4).  Stock torch.nn; torchvision-compatible topology."""
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, stride=2, padding=1)]
    c_in = 64
    for width, blocks, stride in ((64, 3, 1), (128, 8, 2), (256, 36, 2), (512, 3, 2)):
        for i in range(blocks):
            st = stride if i == 0 else 1
            ds = None
            if st != 1 or c_in != 4 * width:
                ds = nn.Sequential(nn.Conv2d(c_in, 4 * width, 1, stride=st, bias=False), nn.BatchNorm2d(4 * width))
            mods.append(_Bottleneck(c_in, width, st, ds))
            c_in = 4 * width
    return nn.Sequential(*mods)


class ImageResNetCoAttentionEncoder(nn.Module):
    """BASELINE config 4 feeder (an extension; the reference only ships the VGG encoder): frozen
    ResNet-152 trunk -> spatial grid [B, N, 2048] as a permuted view of [B, 2048, N]."""

    def __init__(self, is_trainable=False, weights_path=None):
        super().__init__()
        self.is_trainable, self.weights_path = is_trainable, weights_path
        self.resnet_encoder = resnet152_features()
        self.flatten = nn.Flatten(start_dim=2, end_dim=3)
        if not is_trainable:
            for prm in self.resnet_encoder.parameters():
                prm.requires_grad = False

    def forward(self, x_img):
        return self.flatten(self.resnet_encoder(x_img)).permute(0, 2, 1)


class PhraseConvPool(nn.Module):
    """1/2/3-gram Conv1d + tanh, then a max over 3 CONSECUTIVE channels of the concatenated
    [uni|bi|tri] vector -- the reference's reshape (model.py:324-332) groups channels 3e..3e+2,
    not the three n-gram responses of one channel; reproduced as is."""

    def __init__(self, emb_dim):
        super().__init__()
        self.conv_unigram = nn.Sequential(nn.ConstantPad1d((0, 0), 0), nn.Conv1d(emb_dim, emb_dim, 1, 1), nn.Tanh())
        self.conv_bigram = nn.Sequential(nn.ConstantPad1d((1, 0), 0), nn.Conv1d(emb_dim, emb_dim, 2, 1), nn.Tanh())
        self.conv_trigram = nn.Sequential(nn.ConstantPad1d((1, 1), 0), nn.Conv1d(emb_dim, emb_dim, 3, 1), nn.Tanh())
        self.max_pool = nn.MaxPool2d(kernel_size=(1, 3))
        self.fast_products = False     # tolerance mode of the HIP path's fp32 products (train.Trainer(precision="fast") sets it)

    def forward(self, x_question):
        if x_question.is_cuda and os.environ.get("VQA_PHRASE_IMPL", "hip") != "stock":
            # MI355X path (csrc/phrase.hip): same parameters, same values; stock modules below on CPU
            from .phrase import phrase_conv_pool
            u, b, t = self.conv_unigram[1], self.conv_bigram[1], self.conv_trigram[1]
            return phrase_conv_pool(x_question, u.weight, u.bias, b.weight, b.bias, t.weight, t.bias,
                                    fast=self.fast_products)
        B, T, E = x_question.shape
        x = x_question.permute(0, 2, 1)                                          # [B, E, T]
        grams = torch.cat([self.conv_unigram(x), self.conv_bigram(x), self.conv_trigram(x)], dim=1)
        grams = grams.permute(0, 2, 1).reshape(B, T, E, 3)                       # groups of 3 consecutive channels
        return self.max_pool(grams).squeeze(dim=3)                               # [B, T, E]


class QuestionCoAttentionEncoder(nn.Module):
    """word (embedding) / phrase (n-gram conv + pool) / sentence (LSTM) features, each [B,T,hidden];
    rows past a question's length are exact zeros at all three levels (padding_idx=0, packed LSTM)."""

    def __init__(self, vocab_size, word_emb_dim, hidden_dim):
        super().__init__()
        self.vocab_size, self.embedding_dim, self.hidden_dim = vocab_size, word_emb_dim, hidden_dim
        self.word_embedding = nn.Embedding(vocab_size, word_emb_dim, padding_idx=0)
        self.phrase_conv_pool = PhraseConvPool(word_emb_dim)
        self.sentence_lstm = nn.LSTM(word_emb_dim, hidden_dim)

    def forward(self, x, x_lens):
        T = x.shape[1]
        lens = x_lens.cpu() if torch.is_tensor(x_lens) else x_lens    # pack_padded_sequence wants CPU lengths
        words = self.word_embedding(x)
        phrases = pack_padded_sequence(self.phrase_conv_pool(words), lens, batch_first=True)
        sentence, _ = self.sentence_lstm(phrases)
        phrases = pad_packed_sequence(phrases, batch_first=True, total_length=T)[0]
        sentence = pad_packed_sequence(sentence, batch_first=True, total_length=T)[0]
        return words, phrases, sentence


class QuestionBertCoAttentionEncoder(nn.Module):
    """BASELINE config 5 (an extension: the reference lists BERT as a TODO, README.md:137-141, and its
    ``--model bert`` flag has no registry entry): the word level of the hierarchy comes from frozen
    768-d token embeddings of a BERT-base encoder, projected to ``hidden_dim``; phrase and sentence
    levels are built on top of it exactly as in QuestionCoAttentionEncoder.  ``bert`` is any module
    mapping token ids [B,T] to [B,T,bert_dim] (e.g. ``transformers.BertModel(...).embeddings``); it is
    frozen.  Pad rows are re-zeroed so that the three levels keep the reference's zero-pad invariant."""

    def __init__(self, bert: nn.Module, bert_dim, hidden_dim):
        super().__init__()
        self.bert = bert
        for prm in self.bert.parameters():
            prm.requires_grad = False
        self.word_proj = nn.Linear(bert_dim, hidden_dim)
        self.phrase_conv_pool = PhraseConvPool(hidden_dim)
        self.sentence_lstm = nn.LSTM(hidden_dim, hidden_dim)
        self.hidden_dim = hidden_dim
        self.bert.eval()

    def train(self, mode: bool = True):
        """The frozen token-embedding module stays in eval mode (its dropout would make the "frozen embeddings"
        of BASELINE config 5 random)."""
        super().train(mode)
        self.bert.eval()
        return self

    def forward(self, x, x_lens):
        T = x.shape[1]
        lens = x_lens.cpu() if torch.is_tensor(x_lens) else x_lens
        with torch.no_grad():
            tok = self.bert(x)
            tok = tok[0] if isinstance(tok, (tuple, list)) else getattr(tok, "last_hidden_state", tok)
        words = self.word_proj(tok) * (x != 0).unsqueeze(-1).to(tok.dtype)     # zero the pad rows
        phrases = pack_padded_sequence(self.phrase_conv_pool(words), lens, batch_first=True)
        sentence, _ = self.sentence_lstm(phrases)
        phrases = pad_packed_sequence(phrases, batch_first=True, total_length=T)[0]
        sentence = pad_packed_sequence(sentence, batch_first=True, total_length=T)[0]
        return words, phrases, sentence


class MLPClassifier(nn.Module):
    """Recursive word -> phrase -> sentence answer head (model.py:400-434): same constructor, submodules and
    state_dict keys as the reference class.  CUDA fp32 tensors take the HIP head (``head.answer_head``: adds,
    concatenations, bias + tanh folded into tile products on the exact-fp32 MFMA); CPU tensors, or
    ``VQA_HEAD_IMPL=stock``, the stock ``nn.Linear`` modules."""

    def __init__(self, hidden_dim, mlp_dim, K):
        super().__init__()
        self.W_w = nn.Linear(hidden_dim, hidden_dim)
        self.W_p = nn.Linear(2 * hidden_dim, hidden_dim)
        self.W_s = nn.Linear(2 * hidden_dim, mlp_dim)
        self.W_h = nn.Linear(mlp_dim, K)
        self.bf16_products = False       # reduced-precision mode of the HIP head (set with --opt_lvl >= 1, train.Trainer)

    def _hip(self, x):
        x0 = x if torch.is_tensor(x) else x[0]
        return x0.is_cuda and os.environ.get("VQA_HEAD_IMPL", "hip") != "stock"

    def _params(self):
        return (self.W_w.weight, self.W_w.bias, self.W_p.weight, self.W_p.bias, self.W_s.weight, self.W_s.bias,
                self.W_h.weight, self.W_h.bias)

    def forward(self, x_img_feats, x_ques_feats):
        if self._hip(x_img_feats):
            return answer_head(x_img_feats, x_ques_feats, *self._params(), bf16=self.bf16_products)
        (q_w, q_p, q_s), (v_w, v_p, v_s) = x_ques_feats, x_img_feats
        h_w = torch.tanh(self.W_w(q_w + v_w))
        h_p = torch.tanh(self.W_p(torch.cat([q_p + v_p, h_w], dim=1)))
        h_s = torch.tanh(self.W_s(torch.cat([q_s + v_s, h_p], dim=1)))
        return self.W_h(h_s)

    def forward_loss(self, x_img_feats, x_ques_feats, labels):
        """(logits, nn.CrossEntropyLoss()(logits, labels)) -- main.py:211 + :214 -- in one call of the HIP head."""
        if self._hip(x_img_feats):
            return answer_head(x_img_feats, x_ques_feats, *self._params(), labels=labels, bf16=self.bf16_products)
        logits = self.forward(x_img_feats, x_ques_feats)
        return logits, F.cross_entropy(logits.float(), labels)


class HierarchicalCoAttentionNet(nn.Module):
    """question encoder -> image encoder -> parallel co-attention (HIP) -> MLP logits [B, K]."""

    def __init__(self, ques_enc_params, img_enc_params, K, mlp_dim=1024):
        super().__init__()
        self.hidden_dim = ques_enc_params["hidden_dim"]
        img_enc_params = dict(img_enc_params)
        if img_enc_params.pop("arch", "vgg11_bn") == "resnet152":     # config 4 extension (2048-channel grid)
            self.image_encoder = ImageResNetCoAttentionEncoder(**img_enc_params)
        else:
            self.image_encoder = ImageCoAttentionEncoder(**img_enc_params)
        if "bert" in ques_enc_params:                 # config 5 extension; the reference has only the LSTM path
            self.question_encoder = QuestionBertCoAttentionEncoder(**ques_enc_params)
        else:
            self.question_encoder = QuestionCoAttentionEncoder(**ques_enc_params)
        self.co_attention = ParallelCoAttention(self.hidden_dim)
        self.mlp_classify = MLPClassifier(self.hidden_dim, mlp_dim, K)
        self.hot_path_graph = False                   # opt-in: replay the hot path from a captured HIP graph (graph.py)
        # train.Trainer's default on CUDA: the hot path as ONE autograd node over static buffers, its C-ABI calls issued
        # eagerly (graph.HotPathGraph(capture=False)); hot_path_direct_grads: the static gradient buffers become param.grad
        self.hot_path_static = False
        self.hot_path_direct_grads = False
        self._graphs = {}

    def forward(self, x_img, x_ques, x_ques_lens):
        return self.forward_features(self.image_encoder(x_img), x_ques, x_ques_lens)

    def forward_features(self, x_img_features, x_ques, x_ques_lens, labels=None):
        """The forward pass from already-encoded image features [B,N,d] (model.py:171-187 minus the
        image encoder call): lets a frozen encoder run ahead on its own stream (train.Trainer).
        `x_img_features` may be a zero-argument callable returning the features.  With `labels` (int64 [B]) the
        mean cross entropy of main.py:214 comes out of the answer head's own call: returns (logits, loss)."""
        x_ques_features = list(self.question_encoder(x_ques, x_ques_lens))
        if callable(x_img_features):            # resolved only now: the question side is queued first
            x_img_features = x_img_features()
        if (labels is not None and (self.hot_path_graph or self.hot_path_static) and x_img_features.is_cuda
                and torch.is_grad_enabled()):
            return self._graphed(x_img_features, x_ques_features, labels)
        x_img_attn, x_ques_attn = self.co_attention(x_img_features, x_ques_features)
        if labels is not None:
            return self.mlp_classify.forward_loss(x_img_attn, x_ques_attn, labels)
        return self.mlp_classify(x_img_attn, x_ques_attn)

    def _graphed(self, x_img_features, x_ques_features, labels):
        """co-attention + answer head + loss, forward AND backward, as one autograd node over static buffers (graph.py),
        built per (B, N, T) on first use: replayed from captured HIP graphs (``net.hot_path_graph = True``,
        ``Trainer(graph=True)``) or with its four C-ABI calls issued eagerly (``net.hot_path_static``, the Trainer's
        default)."""
        from . import _lib
        from .graph import HotPathGraph
        B, N, _ = x_img_features.shape
        T = x_ques_features[0].shape[1]
        key = (B, N, T, bool(x_img_features.requires_grad), bool(self.co_attention.bf16_projections),
               bool(self.mlp_classify.bf16_products), bool(self.hot_path_graph), bool(self.hot_path_direct_grads),
               bool(self.co_attention.fast_products))
        hp = self._graphs.get(key)
        # (the node reads the parameters where they lie: one built before the module was moved -- .to(), .cuda(), new
        #  Parameter objects -- would read the old storage)
        if hp is not None and hp.stale():          # (all 16 parameters: any of them may have been replaced or re-pointed -- ADVICE r4)
            hp = None
        if hp is None:
            hp = self._graphs[key] = HotPathGraph(self.co_attention, self.mlp_classify, B, N, T, need_dv=key[3],
                                                  flags=(_lib.FLAG_BF16_PROJ if key[4] else 0) | _lib.precision_flag(key[8] and not key[4]),
                                                  capture=key[6], direct_grads=key[7])
        return hp(x_img_features, x_ques_features, labels)


# ---- baseline model (BASELINE config 1: CPU plumbing, no custom kernels) ------------------
class ImageBaselineEncoder(nn.Module):
    """VGG11-bn up to fc7 (4096), L2-normalised, -> 1024 tanh."""

    def __init__(self, is_trainable, weights_path):
        super().__init__()
        self.is_trainable, self.weights_path = is_trainable, weights_path
        features = vgg11_bn_features()
        head = nn.Sequential(*_vgg11_bn_classifier_head())
        _load_vgg_weights(features, head, weights_path)
        self.vgg11_encoder = nn.Sequential(OrderedDict([
            ("conv_layers", features), ("avgpool", nn.AdaptiveAvgPool2d((7, 7))),
            ("fc_layers", nn.Sequential(nn.Flatten(), *list(head)))]))
        self.embedding_layer = nn.Sequential(nn.Linear(4096, 1024), nn.Tanh())
        if not is_trainable:
            for prm in self.vgg11_encoder.parameters():
                prm.requires_grad = False

    def forward(self, x_img):
        return self.embedding_layer(F.normalize(self.vgg11_encoder(x_img), dim=1, p=2))


class QuestionBaselineEncoder(nn.Module):
    """Embedding + tanh -> GRU (last non-pad state via packing) -> 1024 tanh."""

    def __init__(self, vocab_size, word_emb_dim, hidden_dim):
        super().__init__()
        self.hidden_dim, self.vocab_size, self.word_emb_dim = hidden_dim, vocab_size, word_emb_dim
        self.word_embedding = nn.Sequential(nn.Embedding(vocab_size, word_emb_dim), nn.Tanh())
        self.gru = nn.GRU(word_emb_dim, hidden_dim)
        self.embedding_layer = nn.Sequential(nn.Linear(hidden_dim, 1024), nn.Tanh())

    def forward(self, x, seq_lengths):
        lens = seq_lengths.cpu() if torch.is_tensor(seq_lengths) else seq_lengths
        packed = pack_padded_sequence(self.word_embedding(x), lens, batch_first=True)
        _, hidden = self.gru(packed)
        return self.embedding_layer(torch.squeeze(hidden, dim=0))


class VQABaselineNet(nn.Module):
    """Element-wise product of image and question embeddings -> MLP -> logits."""

    def __init__(self, ques_enc_params, img_enc_params, K):
        super().__init__()
        self.image_encoder = ImageBaselineEncoder(**img_enc_params)
        self.question_encoder = QuestionBaselineEncoder(**ques_enc_params)
        self.mlp = nn.Sequential(nn.Linear(1024, 1000), nn.Dropout(0.5), nn.Tanh())
        self.fc_final = nn.Linear(1000, K)

    def forward(self, x_img, x_ques, x_ques_len):
        joint = self.image_encoder(x_img) * self.question_encoder(x_ques, x_ques_len)
        return self.fc_final(self.mlp(joint))
