"""Training loop of the reference (``main.py``) restated for MI355X: same ``--model attention``
registry, same step order (sort batch by length -> H2D -> forward -> CrossEntropyLoss ->
zero_grad -> backward -> Adam step; main.py:193-222), plus what the reference lacks: synthetic
data (no dataset / network here) and data-parallel training, one process per GPU, with RCCL
gradient all-reduce (``vqa_amd.dist``).  apex AMP (main.py:185) is CUDA-only; ``--opt_lvl 0``
(fp32) is the parity mode, levels >= 1 map to ``torch.autocast(bfloat16)`` around the stock
encoders while the co-attention path keeps computing in fp32.

Run:  python -m torch.distributed.run --nproc-per-node N -m vqa_amd.train --synthetic ...
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import time
from typing import Dict

import torch
import torch.nn as nn
import torch.utils.data

from . import dist as vdist
from .loss import CrossEntropyLoss
from .modules import HierarchicalCoAttentionNet, VQABaselineNet

PATH_VGG_WEIGHTS = None      # the reference hard-codes a local .pth (utils.py:15); none ships here


def usable_cpus() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def str2bool(v):
    return str(v).lower() in ("yes", "true", "t", "1")


def int_min_two(k):
    k = int(k)
    if k < 2:
        raise argparse.ArgumentTypeError("must be >= 2")
    return k


def setup_model_configs(args, vocab_size: int, vgg_train: bool = False, vgg_wts_path=None) -> dict:
    """--model -> class, image size and encoder parameters (main.py:388-418).  Called as the reference calls it,
    ``setup_model_configs(args, vocab_size)`` with the parsed command line (main.py:388: ``args.model``,
    ``args.vgg_train``, ``args.vgg_wts_path`` are read), or with the model name and the two VGG settings spelled out.
    The 'attention' entry's ``mlp_dim`` key is carried but never forwarded by the reference (main.py:164), so the
    model's own default (1024, model.py:160) applies; both are 1024."""
    if isinstance(args, str):
        model_name = args
    else:                                            # the reference's argparse namespace
        model_name = args.model
        vgg_train = getattr(args, "vgg_train", vgg_train)
        vgg_wts_path = getattr(args, "vgg_wts_path", vgg_wts_path)
    img = dict(is_trainable=vgg_train, weights_path=vgg_wts_path or PATH_VGG_WEIGHTS)
    if model_name == "attention_bert":
        # BASELINE config 5 (extension; the reference lists BERT as a TODO, README.md:137-141): the attention model with
        # frozen BERT-base token embeddings (768-d, WordPiece vocabulary of 30,522; built from the config -- random init,
        # there is no network for checkpoints) projected to the hidden size as the word level.  `vocab_size` is BERT's own.
        return dict(model=HierarchicalCoAttentionNet, image_size=(448, 448), image_params=img,
                    question_params=bert_question_params(hidden_dim=512), mlp_dim=1024, vocab_size=BERT_VOCAB)
    registry = {
        "baseline": dict(model=VQABaselineNet, image_size=(224, 224), image_params=img,
                         question_params=dict(vocab_size=vocab_size, word_emb_dim=300, hidden_dim=1024)),
        "attention": dict(model=HierarchicalCoAttentionNet, image_size=(448, 448), image_params=img,
                          question_params=dict(vocab_size=vocab_size, word_emb_dim=512, hidden_dim=512),
                          mlp_dim=1024),
        # BASELINE config 4 (extension): ResNet-152 7x7x2048 grid, hidden size 2048; run with --opt_lvl 1
        # for bf16 autocast around the stock encoders and the bf16-MFMA projections of the co-attention
        "attention_resnet": dict(model=HierarchicalCoAttentionNet, image_size=(224, 224),
                                 image_params=dict(img, arch="resnet152"),
                                 question_params=dict(vocab_size=vocab_size, word_emb_dim=2048, hidden_dim=2048),
                                 mlp_dim=1024),
    }
    return registry[model_name]      # 'bert' is accepted by the reference's argparse but has no entry: KeyError


BERT_VOCAB = 30522                   # bert-base-uncased's WordPiece vocabulary


def bert_question_params(hidden_dim: int = 512, vocab_size: int = BERT_VOCAB, bert_dim: int = 768) -> dict:
    """Question-encoder parameters for BASELINE config 5: frozen BERT-base token embeddings (random
    init -- no network for checkpoints) in place of the learned word embedding."""
    from transformers import BertConfig
    from transformers.models.bert.modeling_bert import BertEmbeddings
    cfg = BertConfig(vocab_size=vocab_size, hidden_size=bert_dim)
    return dict(bert=BertEmbeddings(cfg), bert_dim=bert_dim, hidden_dim=hidden_dim)


def build_model(model_name: str, vocab_size: int, num_cls: int, **kw) -> nn.Module:
    """K + 1 output classes: index 0 is UNKNOWN (main.py:155)."""
    cfg = setup_model_configs(model_name, vocab_size, **kw)
    return cfg["model"](cfg["question_params"], cfg["image_params"], K=num_cls + 1)


def sort_batch(images, questions, answers, ques_seq_lens):
    """Descending by question length, as packing requires (utils.py:33-45)."""
    ques_seq_lens, order = ques_seq_lens.sort(dim=0, descending=True)
    return images[order], questions[order], answers[order], ques_seq_lens


def synthetic_batch(batch_size: int, image_size, max_seq_len: int, vocab_size: int, num_classes: int,
                    seed: int) -> Dict[str, torch.Tensor]:
    """One batch shaped like VQADataset's output (dataloader.py:72): images N(0,1), token ids
    U{2..vocab-1} zero-padded to max_seq_len, lengths U{3..max} with at least one full-length
    question, labels U{0..num_classes-1}.  CPU tensors; lengths are NOT sorted (sort_batch does)."""
    g = torch.Generator().manual_seed(seed)
    H, W = image_size
    image = torch.randn(batch_size, 3, H, W, generator=g)
    lens = torch.randint(min(3, max_seq_len), max_seq_len + 1, (batch_size,), generator=g)
    lens[int(torch.randint(0, batch_size, (1,), generator=g))] = max_seq_len
    question = torch.randint(2, vocab_size, (batch_size, max_seq_len), generator=g)
    question = question * (torch.arange(max_seq_len)[None, :] < lens[:, None])
    label = torch.randint(0, num_classes, (batch_size,), generator=g)
    return {"image": image, "question": question, "ques_len": lens, "label": label}


class DevicePrefetcher:
    """Host -> device hand-off of the following batches while the current step computes
    (main.py:205-208 copies synchronously).  A worker thread pulls batches from the host iterator,
    stages them in pinned buffers (a ring of two sets, reused once their copy has completed) and queues
    the copies on a side stream; the consumer stream waits on the copy's event before it touches the
    tensors.  Lengths stay on the host (packing).  At most `depth` batches are in flight."""

    _END = object()

    def __init__(self, batches, device, channels_last: bool = False, depth: int = 2):
        self.device = device
        self.cl = channels_last
        self.next = None
        if device.type != "cuda":
            self._it = iter(batches)
            self._q = None
            self._advance()
            return
        import queue
        import threading
        self.stream = torch.cuda.Stream(device)
        self._q = queue.Queue(maxsize=max(1, depth))
        self._err = None
        self._thread = threading.Thread(target=self._worker, args=(iter(batches),), daemon=True)
        self._thread.start()
        self._advance()

    def _worker(self, it):
        try:
            torch.cuda.set_device(self.device)
            ring = [{}, {}]                                       # pinned staging buffers + their last copy event
            for i, (image, question, ques_len, label) in enumerate(it):
                slot = ring[i & 1]
                if slot.get("event") is not None:
                    slot["event"].synchronize()                  # the previous copy out of this set is done

                def pin(name, t):
                    buf = slot.get(name)
                    if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                        buf = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                        slot[name] = buf
                    buf.copy_(t)
                    return buf

                with torch.cuda.stream(self.stream):
                    im = pin("image", image).to(self.device, non_blocking=True)
                    if self.cl:
                        im = im.contiguous(memory_format=torch.channels_last)
                    qu = pin("question", question).to(self.device, non_blocking=True)
                    la = pin("label", label).to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                slot["event"] = ev
                self._q.put((im, qu, ques_len, la, ev))
        except BaseException as e:                                # surfaced in the consumer thread
            self._err = e
        finally:
            self._q.put(self._END)

    def _advance(self):
        if self._q is None:
            try:
                image, question, ques_len, label = next(self._it)
                self.next = (image, question, ques_len, label, None)
            except StopIteration:
                self.next = None
            return
        item = self._q.get()
        if item is self._END:
            self.next = None
            if self._err is not None:
                raise self._err
        else:
            self.next = item

    def __iter__(self):
        return self

    def peek_image(self):
        """(image, copy-done event) of the batch the next __next__ will return, (None, None) at the
        end: lets the consumer queue work on that image on another stream behind the event."""
        if self.next is None:
            return None, None
        return self.next[0], self.next[4]

    def __next__(self):
        if self.next is None:
            raise StopIteration
        im, qu, ln, la, ev = self.next
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (im, qu, la):
                t.record_stream(cur)
        self._advance()
        return im, qu, ln, la


class Trainer:
    """Owns model, Adam (lr, PyTorch defaults: main.py:180), loss and the gradient reducer.

    Encoder run-ahead: when the image encoder is frozen (the reference default, main.py:67 /
    model.py:239-241) its output does not depend on the optimiser state, so ``step(...,
    next_image=...)`` launches the stock encoder for the NEXT batch on a second, high-priority HIP
    stream before it queues this step's own work.  The long MIOpen convolutions then overlap the
    many short, latency-bound kernels of the question encoder / co-attention / MLP / backward / Adam
    (and the RCCL all-reduce) instead of running in series with them.  Values are unchanged: the same
    modules see the same tensors, BatchNorm running statistics are updated in batch order on that
    stream."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, device=None, opt_lvl: int = 0,
                 bucket_mb: float = 16.0, encoder_runahead: bool = True, graph: bool = False, static_hot_path: bool = True,
                 precision: str = "exact"):
        self.device = device or next(model.parameters()).device
        self.model = model
        # Precision of the HIP path's fp32 products (include/coattn.h "Widths of the fp32 mode"): "exact" -- the default, the
        # reference's arithmetic (main.py --opt_lvl 0) -- is fp32-accurate products over fp32's range, the modules' and the
        # C-ABI's own default; "fast" is the opt-in tolerance mode (forward products on two FP16 pieces = 22 significand
        # bits, backward on two bf16 pieces = 16; inside the reference contract of 1e-4 for operands below 65,504 in
        # magnitude; every forward there folds its range report into a sticky accumulator, and check_range() -- called where
        # the loss is read -- falls back to exact when ANY step since the last check left the range).
        if precision not in ("fast", "exact"):
            raise ValueError("precision must be 'fast' or 'exact'")
        self.set_precision(precision)
        self.criterion = CrossEntropyLoss()      # nn.CrossEntropyLoss() semantics (main.py:94); fused HIP kernel on CUDA
        self.optimizer = torch.optim.Adam(model.parameters(), lr)
        self.opt_lvl = opt_lvl
        if opt_lvl > 0 and hasattr(model, "co_attention"):      # AMP: projections on the bf16 MFMA as well
            model.co_attention.bf16_projections = True
            if hasattr(getattr(model, "mlp_classify", None), "bf16_products"):
                model.mlp_classify.bf16_products = True
        # graph=True: co-attention + answer head + loss, forward and backward, replayed from one captured HIP graph
        # (graph.py): one host call instead of ~25 launches; same values bit for bit
        if graph and hasattr(model, "hot_path_graph") and self.device.type == "cuda":
            model.hot_path_graph = True
        self.reducer = vdist.GradReducer(model, bucket_mb=bucket_mb) if vdist.world_size() > 1 else None
        # Default on CUDA: the hot path as one autograd node over static buffers, calls issued eagerly (graph.py,
        # capture=False): half the host time of the module-by-module path, no graph-node gaps.  This trainer owns the step
        # (zero_grad -> backward -> optimiser), so the static gradient buffers may become param.grad directly -- unless a
        # gradient reducer needs autograd's post-accumulate hooks to fire (data-parallel runs).
        # VQA_HOT_PATH=modules restores the module-by-module path.
        if (static_hot_path and hasattr(model, "hot_path_static") and self.device.type == "cuda"
                and os.environ.get("VQA_HOT_PATH", "static") != "modules"):
            model.hot_path_static = True
            model.hot_path_direct_grads = self.reducer is None
        enc = getattr(model, "image_encoder", None)
        self.runahead = bool(encoder_runahead and self.device.type == "cuda" and enc is not None
                             and hasattr(model, "forward_features")
                             and not any(p.requires_grad for p in enc.parameters()))
        self.enc_stream = (torch.cuda.Stream(self.device, priority=int(os.environ.get("VQA_ENC_PRIORITY", "-1")))
                           if self.runahead else None)
        self._ahead = None                                       # (image, features, event) of the next batch
        self._resident = None                                    # last image batch consumed on the device

    def _autocast(self):
        if self.opt_lvl > 0 and self.device.type == "cuda":
            return torch.autocast("cuda", dtype=torch.bfloat16)
        return contextlib.nullcontext()

    def _queue_encoder(self, image, ready=None):
        """Queue the frozen image encoder for `image` on the encoder stream; returns (image, features,
        done-event).  `ready`: event after which `image` is valid (None: valid once the work queued on
        the current stream so far is done; nothing to wait for if the batch is the one just consumed).
        The encoder stream takes no other dependency on the step, so consecutive passes run back to back."""
        if ready is not None:
            self.enc_stream.wait_event(ready)
        elif image is not self._resident:
            self.enc_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.enc_stream), torch.no_grad(), self._autocast():
            feats = self.model.image_encoder(image)
            ev = torch.cuda.Event()
            ev.record(self.enc_stream)
        image.record_stream(self.enc_stream)
        return image, feats, ev

    def _claim(self, queued):
        """Hand features queued on the encoder stream to the current stream (which waits for them here)."""
        _, feats, ev = queued
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        feats.record_stream(main)
        return feats

    def step(self, image, question, ques_len, label, next_image=None, next_ready=None) -> torch.Tensor:
        """One optimisation step on device-resident, length-sorted tensors; returns the loss.
        `next_image`: the following step's image batch (device-resident), if known; `next_ready`: the
        event that marks its host->device copy complete (DevicePrefetcher.peek_image)."""
        if self.runahead:
            with self._autocast():
                mine, self._ahead = self._ahead, None
                if mine is not None and mine[0] is not image:
                    # queued for some other batch: drop it -- but its pass may still be running on the encoder stream
                    # and updates the same BatchNorm running statistics as the inline pass below: wait for it
                    torch.cuda.current_stream(self.device).wait_stream(self.enc_stream)
                    mine = None
                if mine is None:
                    with torch.no_grad():
                        inline = self.model.image_encoder(image)
                    # the inline pass updated the BatchNorm running statistics on this stream: the pass queued
                    # below for the next batch must come after it (batch order)
                    self.enc_stream.wait_stream(torch.cuda.current_stream(self.device))
                self._resident = image
                if next_image is not None:
                    self._ahead = self._queue_encoder(next_image, next_ready)
                # the question side is queued before the current stream waits for the image features; the loss comes
                # out of the answer head's own call (main.py:211 + :214 in one)
                logits, loss = self.model.forward_features((lambda: self._claim(mine)) if mine is not None else inline,
                                                           question, ques_len, labels=label)
        else:
            with self._autocast():
                logits = self.model(image, question, ques_len)
            loss = self.criterion(logits.float(), label)
        self.optimizer.zero_grad()
        if self.reducer is not None:
            self.reducer.prepare()
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        return loss

    def set_precision(self, precision: str) -> None:
        self.precision = precision
        fast = precision == "fast"
        co = getattr(self.model, "co_attention", None)
        if co is not None and hasattr(co, "fast_products"):
            co.fast_products = fast
        for m in self.model.modules():
            if m is not co and hasattr(m, "fast_products"):
                m.fast_products = fast

    def check_range(self) -> bool:
        """Tolerance mode only: True if every operand of EVERY forward since the last check lay inside the FP16-piece range
        (the report is sticky: each tolerance-mode forward folds its status words into a per-device accumulator,
        _lib.fold_range).  If one did not (its pieces were clamped: that step's values were off, and the Adam updates made
        since are NOT rolled back), warn, switch this trainer to the exact mode for the following steps and return False.
        Data-parallel runs decide together: the flag is MAX-reduced over the group, so every rank switches in the same
        step.  Synchronises -- call it where the loss is read on the host."""
        if self.device.type != "cuda" or self.precision != "fast" or self.opt_lvl > 0:
            return True
        from . import _lib
        msg = None
        try:
            _lib.check_range()
        except _lib.RangeError as e:
            msg = str(e)
        bad = msg is not None
        if vdist.world_size() > 1:
            flag = torch.tensor([1.0 if bad else 0.0], device=self.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if float(flag) > 0 and not bad:
                bad, msg = True, "another rank reported an operand outside the FP16-piece range"
        if bad:
            import warnings
            warnings.warn("vqa_amd.Trainer: %s -- continuing in the exact mode (steps since the last check ran on clamped "
                          "pieces and are not rolled back)" % msg)
            self.set_precision("exact")
            return False
        return True

    def check_labels(self) -> None:
        """Raise IndexError if a label of the last step lay outside [0, K) (the asynchronous HIP loss only sets a status
        word where nn.CrossEntropyLoss raises); synchronises -- call it where the loss is read on the host."""
        if self.device.type == "cuda":
            from . import head, loss
            head.check_labels()
            loss.check_labels()

    @torch.no_grad()
    def validate(self, batches) -> Dict[str, float]:
        """Accuracy / mean CE under eval() (main.py:290-351, without its n_iters+1 / n_iters slip).
        With encoder run-ahead, an encoder pass queued for the next training batch is waited for first
        (its BatchNorm running-statistics update is then already included, i.e. one batch earlier than
        in the serial schedule); pass next_image=None on the step before validating to avoid that."""
        if self.enc_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.enc_stream)
        self.model.eval()
        n_ok = n = 0
        loss = 0.0
        for image, question, ques_len, label in batches:
            logits = self.model(image, question, ques_len)
            loss += float(self.criterion(logits.float(), label))
            n_ok += int((logits.argmax(1) == label).sum())
            n += label.numel()
        self.model.train()
        return {"accuracy": 100.0 * n_ok / max(n, 1), "loss": loss / max(len(batches), 1)}


class SyntheticVQADataset(torch.utils.data.Dataset):
    """Samples shaped like VQADataset's (dataloader.py:72: {'image','question','ques_len','label'}) drawn from a seed per
    index: images N(0,1), token ids U{2..vocab-1} zero-padded to max_seq_len, lengths U{3..max}, labels U{0..K}.  Stands
    in for the dataset files this environment does not have; feeds the same DataLoader(batch_size, shuffle, drop_last,
    num_workers) as main.py:129-130."""

    def __init__(self, n_samples, image_size, max_seq_len, vocab_size, num_classes, seed):
        self.n, self.size, self.T, self.vocab, self.K, self.seed = n_samples, image_size, max_seq_len, vocab_size, num_classes, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        n = int(torch.randint(min(3, self.T), self.T + 1, (1,), generator=g))
        q = torch.randint(2, self.vocab, (self.T,), generator=g) * (torch.arange(self.T) < n)
        return {"image": torch.randn(3, *self.size, generator=g), "question": q, "ques_len": torch.tensor(n),
                "label": torch.randint(0, self.K, (), generator=g)}


def build_parser():
    """The reference's command line (main.py:34-78; same names, types and meanings) plus what it lacks here: synthetic
    data, precision, data-parallel launch through the environment.  The dataset / vocabulary flags are accepted so that
    a reference command line parses; only --synthetic data exists in this environment."""
    ap = argparse.ArgumentParser(description="Visual Question Answering (MI355X co-attention path)")
    # Experiment params (main.py:37-41; not `required` here: synthetic runs need no directory)
    ap.add_argument("--mode", default="train", choices=["train", "test"])
    ap.add_argument("--expt_dir", type=str, default=None, help="root directory to save model & summaries")
    ap.add_argument("--expt_name", type=str, default="expt")
    ap.add_argument("--run_name", type=str, default="run")
    ap.add_argument("--model", default="attention", choices=["baseline", "attention", "bert", "attention_resnet", "attention_bert"])
    # Data params (main.py:44-48)
    ap.add_argument("--train_img", type=str, default=None)
    ap.add_argument("--train_file", type=str, default=None)
    ap.add_argument("--val_img", type=str, default=None)
    ap.add_argument("--val_file", type=str, default=None)
    ap.add_argument("--num_cls", "-K", type=int_min_two, default=1000)
    ap.add_argument("--vocab_file", type=str, default=None)
    # Training params (main.py:54-59)
    ap.add_argument("--batch_size", "-bs", type=int, default=8)
    ap.add_argument("--num_epochs", "-ep", type=int, default=1,
                    help="number of epochs (the reference's default is 50; an epoch here is --num_steps synthetic batches)")
    ap.add_argument("--learning_rate", "-lr", type=float, default=1e-4)
    ap.add_argument("--log_interval", type=int, default=10, help="interval size for logging training summaries")
    ap.add_argument("--save_interval", type=int, default=3000, help="save model after `n` weight update steps (main.py:260-263)")
    ap.add_argument("--val_size", type=int, default=0,
                    help="validation set size for evaluating accuracy (samples; the reference's default is 10000; 0 = no "
                         "validation: there is no --val_file here, the samples are synthetic)")
    ap.add_argument("--K_eval", type=int, default=1000)
    # Model params (main.py:65-67)
    ap.add_argument("--model_ckpt", type=str, default=None, help="resume training; e.g. model_1000.pth (inside the log directory, or a path)")
    ap.add_argument("--vgg_wts_path", type=str, default=None)
    ap.add_argument("--vgg_train", type=str2bool, default="false")
    # GPU params (main.py:72-73)
    ap.add_argument("--gpu_id", type=int, default=0, help="cuda:gpu_id of a single-process run (under a launcher LOCAL_RANK decides)")
    ap.add_argument("--opt_lvl", type=int, default=0, choices=[0, 1, 2, 3],
                    help="0 = fp32 (the parity mode; the reference's default is apex O1, which is CUDA-only); >= 1: bf16 autocast")
    # Misc params (main.py:76)
    ap.add_argument("--num_workers", type=int, default=0, help="number of worker processes of the DataLoader")
    # --- this build ---
    ap.add_argument("--num_steps", type=int, default=20, help="synthetic batches per epoch")
    ap.add_argument("--precision", default="exact", choices=["fast", "exact"],
                    help="fp32 products of the HIP path: fp32-accurate (default: the reference's --opt_lvl 0 arithmetic) or the "
                         "opt-in tolerance mode (include/coattn.h COATTN_FLAG_FAST16; range-checked at every --log_interval)")
    ap.add_argument("--synthetic", type=str2bool, default="true", help="synthetic data (the only source here)")
    ap.add_argument("--vocab_size", type=int, default=10000)
    ap.add_argument("--max_seq_length", type=int, default=26)
    ap.add_argument("--image_size", type=int, default=0, help="override the model's image size (0 = registry)")
    ap.add_argument("--channels_last", type=str2bool, default="true",
                    help="run the stock image encoder in channels_last (MIOpen NHWC kernels)")
    ap.add_argument("--val_interval", type=int, default=0,
                    help="validate every this many steps (0: at --log_interval, as main.py:225-246, when --val_size > 0)")
    ap.add_argument("--val_batches", type=int, default=0, help="validation batches (overrides --val_size // --batch_size)")
    ap.add_argument("--save_path", type=str, default=None, help="also save the final state_dict here")
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "test":
        raise NotImplementedError("TODO: test mode")          # as the reference, main.py:286-287
    if not args.synthetic:
        raise SystemExit("only --synthetic true is available: the VQA dataset is not present in this environment")

    rank, world, local = vdist.init_from_env()
    if torch.cuda.is_available():
        device = torch.device("cuda", local if "LOCAL_RANK" in os.environ else args.gpu_id)   # (main.py:80)
    else:
        device = torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
        # the step is GPU work; the host side only launches kernels and stages batches.  ATen's default of
        # one thread per logical CPU oversubscribes a containerised rank (measured: 39.7 vs 26.3 ms/step)
        torch.set_num_threads(max(1, min(4, usable_cpus())))
    torch.manual_seed(0)                                        # identical weights on every rank
    cfg = setup_model_configs(args, args.vocab_size)             # (as main.py:388)
    args.vocab_size = cfg.get("vocab_size", args.vocab_size)     # attention_bert: token ids are BERT's
    model = cfg["model"](cfg["question_params"], cfg["image_params"], K=args.num_cls + 1)
    # log directory expt_dir/expt_name/run_name (main.py:110-114): checkpoints model_{step}.pth live there (main.py:260-263)
    log_dir = os.path.join(args.expt_dir, args.expt_name, args.run_name) if args.expt_dir else None
    if log_dir and rank == 0:
        os.makedirs(log_dir, exist_ok=True)
    if args.model_ckpt:
        ckpt = args.model_ckpt
        if log_dir and not os.path.isabs(ckpt) and os.path.exists(os.path.join(log_dir, ckpt)):
            ckpt = os.path.join(log_dir, ckpt)                   # (main.py:169)
        model.load_state_dict(torch.load(ckpt, map_location="cpu"))
    model.to(device)
    cl = args.channels_last and device.type == "cuda" and args.model.startswith("attention")
    if cl:
        model.image_encoder.to(memory_format=torch.channels_last)
    trainer = Trainer(model, args.learning_rate, device, args.opt_lvl, precision=args.precision)
    size = (args.image_size, args.image_size) if args.image_size else cfg["image_size"]
    n_cls = args.num_cls + 1

    def loader(n_samples, seed):
        ds = SyntheticVQADataset(n_samples, size, args.max_seq_length, args.vocab_size, n_cls, seed)
        return torch.utils.data.DataLoader(ds, args.batch_size, shuffle=True, drop_last=True, num_workers=args.num_workers,
                                           generator=torch.Generator().manual_seed(seed))

    def sorted_host(batch):                                      # main.py:196-202
        image, question, label, ques_len = sort_batch(batch["image"], batch["question"], batch["label"], batch["ques_len"])
        return image, question, ques_len, label

    train_loader = loader(args.num_steps * args.batch_size, 1234 + rank)
    steps_per_epoch = len(train_loader)
    val = []
    n_val = args.val_batches or args.val_size // args.batch_size
    if n_val > 0:
        for b in loader(n_val * args.batch_size, 987654 + rank):
            image, question, ques_len, label = sorted_host(b)
            image = image.to(device)
            if cl:
                image = image.contiguous(memory_format=torch.channels_last)
            val.append((image, question.to(device), ques_len, label.to(device)))
    val_every = args.val_interval or (args.log_interval if val else 0)

    t0 = time.time()
    step = 0
    for epoch in range(args.num_epochs):
        batches = DevicePrefetcher((sorted_host(b) for b in train_loader), device, cl)
        for image, question, ques_len, label in batches:
            validate_now = bool(val) and val_every > 0 and (step + 1) % val_every == 0
            # no encoder run-ahead across a validation: its BatchNorm statistics must be those of this step
            nxt, ready = (None, None) if validate_now else batches.peek_image()
            loss = trainer.step(image, question, ques_len, label, next_image=nxt, next_ready=ready)
            if (step + 1) % args.log_interval == 0:
                trainer.check_labels()                           # (the host synchronises here anyway to read the loss)
                trainer.check_range()
                if rank == 0:
                    print(json.dumps({"epoch": epoch + 1, "step": step + 1, "steps_per_epoch": steps_per_epoch,
                                      "loss": round(float(loss), 5), "precision": trainer.precision,
                                      "pairs_per_s": round(world * args.batch_size * (step + 1) / (time.time() - t0), 2)}))
            if validate_now:
                m = trainer.validate(val)
                if rank == 0:
                    print(json.dumps({"step": step + 1, "val_accuracy": round(m["accuracy"], 3), "val_loss": round(m["loss"], 5)}))
            if (step + 1) % args.save_interval == 0 and log_dir and rank == 0:   # main.py:260-263
                torch.save(model.state_dict(), os.path.join(log_dir, "model_%d.pth" % (step + 1)))
            step += 1
    if args.save_path and rank == 0:
        torch.save(model.state_dict(), args.save_path)
    vdist.shutdown()


if __name__ == "__main__":
    main()
