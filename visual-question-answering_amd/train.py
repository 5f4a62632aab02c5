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
                 bucket_mb: float = 16.0, encoder_runahead: bool = True, graph: bool = False, static_hot_path: bool = True):
        self.device = device or next(model.parameters()).device
        self.model = model
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


def main(argv=None):
    ap = argparse.ArgumentParser(description="Visual Question Answering (MI355X co-attention path)")
    ap.add_argument("--mode", default="train", choices=["train", "test"])
    ap.add_argument("--model", default="attention", choices=["baseline", "attention", "bert", "attention_resnet", "attention_bert"])
    ap.add_argument("--num_cls", "-K", type=int_min_two, default=1000)
    ap.add_argument("--batch_size", "-bs", type=int, default=8)
    ap.add_argument("--num_steps", type=int, default=20)
    ap.add_argument("--learning_rate", "-lr", type=float, default=1e-4)
    ap.add_argument("--log_interval", type=int, default=10)
    ap.add_argument("--vgg_wts_path", type=str, default=None)
    ap.add_argument("--vgg_train", type=str2bool, default="false")
    ap.add_argument("--opt_lvl", type=int, default=0, choices=[0, 1, 2, 3])
    ap.add_argument("--synthetic", type=str2bool, default="true", help="synthetic data (the only source here)")
    ap.add_argument("--vocab_size", type=int, default=10000)
    ap.add_argument("--max_seq_length", type=int, default=26)
    ap.add_argument("--image_size", type=int, default=0, help="override the model's image size (0 = registry)")
    ap.add_argument("--channels_last", type=str2bool, default="true",
                    help="run the stock image encoder in channels_last (MIOpen NHWC kernels)")
    ap.add_argument("--val_interval", type=int, default=0,
                    help="every this many steps: accuracy / loss under eval() on --val_batches synthetic batches "
                         "(main.py:242-257, :290-351); 0 = off")
    ap.add_argument("--val_batches", type=int, default=2)
    ap.add_argument("--model_ckpt", type=str, default=None)
    ap.add_argument("--save_path", type=str, default=None)
    args = ap.parse_args(argv)
    if args.mode == "test":
        raise NotImplementedError("TODO: test mode")          # as the reference, main.py:286-287
    if not args.synthetic:
        raise SystemExit("only --synthetic true is available: the VQA dataset is not present in this environment")

    rank, world, local = vdist.init_from_env()
    device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
        # the step is GPU work; the host side only launches kernels and stages batches.  ATen's default of
        # one thread per logical CPU oversubscribes a containerised rank (measured: 39.7 vs 26.3 ms/step)
        torch.set_num_threads(max(1, min(4, usable_cpus())))
    torch.manual_seed(0)                                        # identical weights on every rank
    cfg = setup_model_configs(args, args.vocab_size)             # (as main.py:388)
    args.vocab_size = cfg.get("vocab_size", args.vocab_size)     # attention_bert: token ids are BERT's
    model = cfg["model"](cfg["question_params"], cfg["image_params"], K=args.num_cls + 1)
    if args.model_ckpt:
        model.load_state_dict(torch.load(args.model_ckpt, map_location="cpu"))
    model.to(device)
    cl = args.channels_last and device.type == "cuda" and args.model.startswith("attention")
    if cl:
        model.image_encoder.to(memory_format=torch.channels_last)
    trainer = Trainer(model, args.learning_rate, device, args.opt_lvl)
    size = (args.image_size, args.image_size) if args.image_size else cfg["image_size"]
    def host_batches():
        for step in range(args.num_steps):
            b = synthetic_batch(args.batch_size, size, args.max_seq_length, args.vocab_size, args.num_cls + 1,
                                seed=1234 + rank + 1000 * step)
            image, question, label, ques_len = sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
            yield image, question, ques_len, label

    val = []
    if args.val_interval > 0:
        for i in range(args.val_batches):
            b = synthetic_batch(args.batch_size, size, args.max_seq_length, args.vocab_size, args.num_cls + 1,
                                seed=987654 + rank + 1000 * i)
            image, question, label, ques_len = sort_batch(b["image"], b["question"], b["label"], b["ques_len"])
            image = image.to(device)
            if cl:
                image = image.contiguous(memory_format=torch.channels_last)
            val.append((image, question.to(device), ques_len, label.to(device)))

    t0 = time.time()
    batches = DevicePrefetcher(host_batches(), device, cl)
    for step, (image, question, ques_len, label) in enumerate(batches):
        validate_now = bool(val) and (step + 1) % args.val_interval == 0
        # no encoder run-ahead across a validation: its BatchNorm statistics must be those of this step
        nxt, ready = (None, None) if validate_now else batches.peek_image()
        loss = trainer.step(image, question, ques_len, label, next_image=nxt, next_ready=ready)
        if (step + 1) % args.log_interval == 0:
            trainer.check_labels()                               # (the host synchronises here anyway to read the loss)
        if (step + 1) % args.log_interval == 0 and rank == 0:
            print(json.dumps({"step": step + 1, "loss": round(float(loss), 5),
                              "pairs_per_s": round(world * args.batch_size * (step + 1) / (time.time() - t0), 2)}))
        if validate_now:
            m = trainer.validate(val)
            if rank == 0:
                print(json.dumps({"step": step + 1, "val_accuracy": round(m["accuracy"], 3), "val_loss": round(m["loss"], 5)}))
    if args.save_path and rank == 0:
        torch.save(model.state_dict(), args.save_path)
    vdist.shutdown()


if __name__ == "__main__":
    main()
