"""Data-parallel training support: one process per GPU, gradients averaged with bucketed
all-reduce (RCCL over xGMI through ``torch.distributed`` backend "nccl"; "gloo" on CPU).

The reference has no multi-GPU path (commented TODO, main.py:71, :102-106); the co-attention
model shards naturally by QA pair, so the only exchange step is the gradient mean
(SURVEY.md section 8e).  ``GradReducer`` differs from stock DDP where this model needs it:

* ``co_attention.W_b`` is constructed but never used (model.py:347 vs :377), so it never gets a
  gradient; stock DDP (find_unused_parameters=False) fails on step 2.  The reducer learns the
  set of parameters that actually receive gradients on the first step and buckets only those.
* buffers (frozen-VGG BatchNorm statistics, model.py:239-241) are never broadcast.
* buckets are filled in reverse registration order (~ backward order) and each all-reduce is
  launched asynchronously as soon as its bucket is complete, overlapping with the rest of
  backward; xGMI is point-to-point, so buckets are kept large (default 16 MB: the ~49 MB
  gradient of the attention model goes out in 3-4 collectives).
* ``bench.py --gpus N`` times the step under every exchange pattern (``reset``) plus a compute-only leg
  (``exchange="none"``: buckets packed and unpacked, nothing sent) and reports the exposed time of each.
* ``exchange="direct"`` (or ``VQA_GRAD_EXCHANGE=direct``) replaces the ring all-reduce by the
  one-shot pattern that fits a fully connected xGMI node: every rank owns 1/world of a bucket, an all-to-all
  sends shard j of every rank's bucket straight to rank j (all 7 peer links carry S/8 each at once instead of
  2 (w-1)/w S circling one ring), the owner sums the w pieces in rank order, and an all-gather returns the reduced
  shards.  Same values on every rank; opt-in until it has been timed against RCCL's own all-reduce on a
  multi-GPU node (a one-GPU box cannot).
* ``exchange="p2p"``: the same one-shot pattern WITHOUT collectives on the data path -- every rank maps its peers' buckets
  once (HIP IPC handles through ``torch.multiprocessing``'s reductions, exchanged with ``all_gather_object``) and the
  library's two bandwidth kernels read peer memory directly (``csrc/p2p.hip``: ``coattn_p2p_reduce_scatter`` sums shard j
  of all ranks in rank order on rank j, ``coattn_p2p_all_gather`` copies the reduced shards back).  The three phase
  boundaries per step (buckets packed / shards reduced / shards gathered) are one-element all-reduces on the stream under
  RCCL (stream-ordered, the host does not block) and barriers under gloo.  Tested with several ranks sharing ONE GPU
  (``tests/test_gpu_dist.py``); between GPUs it relies on peer access, which ``_p2p_setup`` switches on and checks with a
  copy through torch before any kernel dereferences a peer pointer.
"""
from __future__ import annotations

import logging
import os
from typing import List

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, local_rank); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("VQA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        import datetime
        # (VQA_DIST_TIMEOUT_S: a rank that never arrives ends the others with an error after this long, not after the
        #  backend's own default)
        tmo = datetime.timedelta(seconds=int(os.environ.get("VQA_DIST_TIMEOUT_S", "600")))
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rk, world_size=world, device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rk, world_size=world, timeout=tmo)
    return rk, world, local


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


EXCHANGES = ("allreduce", "direct", "p2p", "none")
TRAINING_EXCHANGES = ("allreduce", "direct", "p2p")    # "none" keeps gradients local: a timing leg (reset()), never a mode to train in
log = logging.getLogger("vqa_amd.dist")


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], pad_to: int = 1):
        self.params = params
        n = sum(p.numel() for p in params)
        self.numel = n
        n = (n + pad_to - 1) // pad_to * pad_to          # direct exchange: equal shards per rank
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.recv = None                                  # direct exchange: the w pieces of this rank's shard
        self.peers = None                                 # p2p exchange: every rank's bucket as mapped here (own: flat)
        self.peer_ptrs = None
        self.views, o = [], 0
        for p in params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.pending = len(params)
        self.work = None
        self.host = None                                  # direct exchange under gloo with CUDA buckets: host copy of the bucket


class GradReducer:
    """Averages ``.grad`` of a module's parameters across ranks.

    Usage per step:  ``prepare()`` -> ``loss.backward()`` -> ``finish()`` -> ``optimizer.step()``.
    """

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 16.0, group=None, exchange: str | None = None):
        self.module = module
        self.group = group
        self.exchange = exchange or os.environ.get("VQA_GRAD_EXCHANGE", "allreduce")
        if self.exchange not in TRAINING_EXCHANGES:
            # ("none" would train on unreduced gradients without a word: it exists for bench.py's compute-only leg and
            #  is reachable through reset() only)
            raise ValueError("exchange must be one of %s, got %r" % (", ".join(TRAINING_EXCHANGES), self.exchange))
        self.fallback_reason = None                    # set when exchange="p2p" could not be set up and all-reduce took over
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.world = dist.get_world_size(group)
        self.buckets: List[_Bucket] | None = None      # built after the first backward
        self._where = {}
        self._hooks = []
        self.unused: List[str] = []

    def reset(self, exchange: str) -> None:
        """Switch the exchange pattern (bench.py times the step under each): hooks and buckets are dropped and rebuilt
        on the next step.  "none" packs and unpacks the buckets but starts no collective -- the gradients stay LOCAL:
        the compute-only leg of a timing comparison, never a training mode."""
        if exchange not in EXCHANGES:
            raise ValueError("exchange must be one of %s, got %r" % (", ".join(EXCHANGES), exchange))
        self._p2p_release()
        for h in self._hooks:
            h.remove()
        self._hooks, self._where, self.buckets = [], {}, None
        self.exchange = exchange

    # -- bucket construction (after the first backward: only parameters that got a gradient) ----
    def _build(self):
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        self.unused = [n for n, p in named if p.grad is None]
        live = [p for _, p in named if p.grad is not None][::-1]      # ~ order gradients become ready
        # every rank must agree on the layout
        sig = torch.tensor([len(live), sum(p.numel() for p in live)], dtype=torch.int64, device=live[0].device)
        lo, hi = sig.clone(), sig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
            raise RuntimeError("ranks disagree on which parameters receive gradients")
        self.buckets, cur, size = [], [], 0
        pad = {"direct": self.world, "p2p": 4 * self.world}.get(self.exchange, 1)   # equal (p2p: 16-byte) shards
        for p in live:
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > self.bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(_Bucket(cur, pad))
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(_Bucket(cur, pad))
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[p] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.exchange == "p2p":
            self._p2p_setup_or_fall_back()

    # -- p2p exchange: peer-mapped buckets ------------------------------------------------------
    def _p2p_setup_or_fall_back(self):
        """exchange="p2p" has only ever run with ranks sharing one GPU.  If mapping a peer or
        switching peer access on fails on ANY rank, every rank drops to RCCL's all-reduce -- the decision is itself
        all-reduced, so the ranks cannot disagree -- and says why once, instead of raising in the middle of a run."""
        err = None
        try:
            self._p2p_setup()
        except Exception as e:                         # noqa: BLE001 -- whatever the mapping raised, the step must go on
            err = "%s: %s" % (type(e).__name__, e)
        bad = torch.tensor([1 if err else 0], dtype=torch.int32, device=self.buckets[0].flat.device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        if int(bad.item()) == 0:
            self._tok = torch.zeros(1, device=self._dev)
            self._p2p_sync()
            return
        for b in self.buckets:
            b.peers, b.peer_ptrs = None, None
        self.fallback_reason = err or "a peer rank could not map this rank's buckets"
        self.exchange = "allreduce"                    # (the buckets' padding is harmless for the all-reduce)
        if dist.get_rank(self.group) == 0 or err:
            log.warning("GradReducer: exchange='p2p' unavailable (%s); using RCCL all-reduce", self.fallback_reason)

    def _p2p_setup(self):
        """Map every peer's buckets into this process (once per bucket layout).  Exactly one collective
        (all_gather_object), which every rank reaches whether or not its own part failed: EVERYTHING that can fail before it
        -- loading the library, the dtype / device checks, the synchronisation, exporting the IPC handles -- runs inside the
        try block whose outcome is what the rank contributes (a rank that raised before the collective left the
        others waiting in it).  A rank that raises afterwards leaves no peer waiting (the caller's consensus all-reduce
        comes next on every rank)."""
        import ctypes as C
        mine = None
        self.rank = dist.get_rank(self.group)
        try:
            from torch.multiprocessing.reductions import reduce_tensor
            from . import _lib
            self._lib = _lib.load()
            for b in self.buckets:
                if b.flat.dtype != torch.float32 or not b.flat.is_cuda:
                    raise RuntimeError("exchange='p2p' takes fp32 gradients on a GPU")
            self._dev = self.buckets[0].flat.device
            torch.cuda.synchronize(self._dev)
            mine = [reduce_tensor(b.flat) for b in self.buckets]           # (rebuild function, IPC handle + offset) per bucket
        except Exception as e:                                             # noqa: BLE001
            mine = "%s: %s" % (type(e).__name__, e)
            self._dev = self.buckets[0].flat.device
        every = [None] * self.world
        dist.all_gather_object(every, mine, group=self.group)
        for r, m in enumerate(every):
            if isinstance(m, str):
                raise RuntimeError("rank %d could not export its buckets (%s)" % (r, m))
        for bi, b in enumerate(self.buckets):
            b.peers = [b.flat if r == self.rank else every[r][bi][0](*every[r][bi][1]) for r in range(self.world)]
            for r, t in enumerate(b.peers):
                if t.numel() != b.flat.numel():
                    raise RuntimeError("rank %d maps a bucket of another size" % r)
                if r != self.rank:
                    # peer access from this rank's device to the bucket's (an error here, not a faulting kernel later),
                    # then one small copy through torch as a check of the mapping
                    with torch.cuda.device(b.flat.device):
                        from . import _lib
                        _lib.check(self._lib.coattn_p2p_enable_peer(t.device.index), "coattn_p2p_enable_peer")
                    b.flat.new_empty(4).copy_(t[:4])
            b.peer_ptrs = (C.c_void_p * self.world)(*[t.data_ptr() for t in b.peers])

    def _p2p_sync(self):
        """Every rank's work enqueued so far is done before any rank's later work starts."""
        if dist.get_backend(self.group) == "nccl":
            with torch.cuda.device(self._dev):
                dist.all_reduce(self._tok, group=self.group)          # on the stream: the host does not block
        else:
            torch.cuda.synchronize(self._dev)
            dist.barrier(group=self.group)

    def _p2p_release(self):
        if self.buckets is None or not any(b.peers is not None for b in self.buckets):
            return
        self._p2p_sync()
        for b in self.buckets:
            b.peers, b.peer_ptrs = None, None                          # unmap before the owners free
        torch.cuda.synchronize(self._dev)
        dist.barrier(group=self.group)

    def _p2p_finish(self):
        from . import _lib
        # (the buckets' own device and its current stream, whatever device the caller has selected: as head.py / loss.py)
        stream = torch.cuda.current_stream(self._dev).cuda_stream
        self._p2p_sync()                                               # every rank's buckets are packed
        with torch.cuda.device(self._dev):
            for b in self.buckets:
                _lib.check(self._lib.coattn_p2p_reduce_scatter(b.peer_ptrs, self.world, self.rank, b.flat.numel() // self.world,
                                                               1.0 / self.world, stream), "coattn_p2p_reduce_scatter")
        self._p2p_sync()                                               # every shard is reduced
        with torch.cuda.device(self._dev):
            for b in self.buckets:
                _lib.check(self._lib.coattn_p2p_all_gather(b.peer_ptrs, self.world, self.rank, b.flat.numel() // self.world,
                                                           stream), "coattn_p2p_all_gather")
        self._p2p_sync()                                               # nobody still reads a bucket that is packed next

    def _launch(self, b):
        """Pack a bucket whose gradients are all there (one multi-tensor copy) and start its all-reduce."""
        torch._foreach_copy_(b.views, [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params])
        if self.exchange in ("none", "p2p"):
            b.work = False                              # packed; p2p: the exchange runs in finish(), "none": nothing sent
        elif self.exchange == "direct":
            # stage 1: shard j of this rank's bucket goes straight to rank j (stage 2 in finish())
            # (gloo has no all-to-all / all-gather on CUDA tensors: there -- ranks rehearsing on one GPU -- the same
            #  collectives run on a host copy of the bucket; the shard arithmetic and the order of the sums are the same)
            src = b.flat
            if self._host_staged(b):
                b.host = b.flat.cpu()
                src = b.host
            if b.recv is None or b.recv.device != src.device:
                b.recv = torch.empty_like(src)
            b.work = dist.all_to_all_single(b.recv, src, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _host_staged(self, b) -> bool:
        return b.flat.is_cuda and dist.get_backend(self.group) == "gloo"

    def _on_grad(self, p):
        bi, _ = self._where[p]
        b = self.buckets[bi]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    # -- per step -----------------------------------------------------------------------------
    def prepare(self):
        if self.buckets is not None:
            for b in self.buckets:
                b.pending, b.work = len(b.params), None

    def finish(self):
        if self.buckets is None:
            # first step: no overlap; discover the live parameter set, reduce synchronously
            self._build()
            for b in self.buckets:
                self._launch(b)
        for b in self.buckets:
            if b.work is None:                          # a parameter of this bucket got no gradient this step
                self._launch(b)
        if self.exchange == "p2p":
            self._p2p_finish()
        if self.exchange == "direct":
            # stage 2: sum the w pieces of the own shard in rank order, scale, gather the reduced shards
            gathers = []
            for b in self.buckets:
                b.work.wait()
                pieces = b.recv.view(self.world, -1)
                shard = pieces[0].clone()
                for r in range(1, self.world):
                    shard.add_(pieces[r])
                shard.div_(self.world)
                dst = b.host if self._host_staged(b) else b.flat
                gathers.append((dist.all_gather_into_tensor(dst, shard, group=self.group, async_op=True), shard))
            for work, _ in gathers:
                work.wait()
            for b in self.buckets:
                if self._host_staged(b):
                    b.flat.copy_(b.host)
        for b in self.buckets:
            if self.exchange == "allreduce":
                b.work.wait()
                b.flat.div_(self.world)
            for v, p in zip(b.views, b.params):
                if p.grad is None:
                    p.grad = torch.empty_like(p)
            torch._foreach_copy_([p.grad for p in b.params], b.views)       # one multi-tensor copy back

    def payload_bytes(self) -> int:
        return sum(b.numel * b.flat.element_size() for b in (self.buckets or []))
