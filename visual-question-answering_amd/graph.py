"""``HotPathGraph``: the fixed-shape launch sequence of the hot path -- co-attention forward + answer head + cross
entropy, and their backward -- captured into HIP graphs and replayed per step (two host calls instead of ~25 kernel
launches and two autograd round trips).

What is captured are the C-ABI calls themselves (``coattn_forward`` + ``coattn_head_forward``; ``coattn_head_backward`` +
``coattn_backward``: asynchronous on the stream they are given, no allocation, no synchronisation -> capture-safe).
Outputs, saved state and workspaces are static buffers of this object; the INPUTS (image features, the three question
levels, labels) are read where they lie: a graph pair is captured per set of input addresses and image-feature strides
(a training loop's allocator hands the same blocks back step after step; at most ``MAX_KEYS`` pairs, beyond that the
inputs are copied into static buffers).  The image features are taken in either physical layout the kernels run on --
contiguous [B,N,d], or the permuted view of the reference's channel-major buffer (model.py:215-217) -- through the
C-ABI's strides, without a copy.  The upstream gradient of the loss is a device scalar the backward graph reads, so any
``loss * k`` upstream works.  Values are bit-for-bit those of the eager C-ABI calls.

    hp = HotPathGraph(co_attention, mlp_classify, B, N, T)          # modules of HierarchicalCoAttentionNet
    logits, loss = hp(x_img, [Q_w, Q_p, Q_s], labels)               # autograd-aware; shapes fixed

``capture=False`` keeps everything above -- static buffers, one autograd node for the whole hot path, argument blocks
built once per address set -- but issues the four C-ABI calls EAGERLY on the current stream instead of replaying
captured graphs: no capture, no graph-node gaps on the device, and a host cost per step close to the replay's (what is
left is the ~25 kernel launches).  It is what ``train.Trainer`` uses by default.  ``direct_grads=True`` (the Trainer sets
it when it owns the step: gradients are consumed by the optimiser before the next backward, and no gradient hooks need
to fire) assigns the static gradient buffers to ``param.grad`` instead of handing them to autograd, whose AccumulateGrad
would clone each of the 16 (it cannot steal a buffer somebody else still holds): 16 copy kernels per step less.
Contract of ``direct_grads``: ``param.grad`` ALIASES a static buffer that the next backward rewrites -- whoever keeps a
reference to it across steps (a gradient logger, an EMA of gradients) must clone it.  A second backward without a
``zero_grad(set_to_none=True)`` in between (micro-batch accumulation) is supported in eager mode: the kernels then ADD into
the buffers (C-ABI ``accumulate = 1``); with captured graphs, or when only some of the 16 parameters still hold the static
buffer, it raises instead of dropping the earlier gradient.

``logits`` is returned as a fresh tensor (``alias_outputs=True``: the static buffer itself, overwritten by the next
step).  A gradient arriving for ``logits`` is added by the head's backward in eager mode; under graph capture it raises.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Sequence

import torch

from . import _lib
from . import head as _head
from .coattention import _is_native, _native_layout, _strides, native_features
# (the package exports a FUNCTION named `coattention`, which shadows the submodule as an attribute: take the module itself)
_coattention = sys.modules[_is_native.__module__]
from .head import _workspace_bytes as _head_ws


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class HotPathGraph:
    MAX_KEYS = 8

    def __init__(self, co_attention, mlp_classify, B: int, N: int, T: int, need_dv: bool = False, flags: int = 0,
                 capture: bool = True, direct_grads: bool = False, alias_outputs: bool = False):
        self.co, self.mlp = co_attention, mlp_classify
        self.capture, self.direct_grads, self.alias_outputs = capture, direct_grads, alias_outputs
        self._warned_accumulate = False                          # (one warning when a backward adds into held gradients)
        d = co_attention.hidden_dim
        mlp, K = mlp_classify.W_s.weight.shape[0], mlp_classify.W_h.weight.shape[0]
        self.dims = (B, N, T, d, mlp, K)
        dev = co_attention.W_v.weight.device
        self.device = dev
        self.flags = flags
        self.head_flags = _lib.FLAG_BF16_PROJ if getattr(mlp_classify, "bf16_products", False) else 0
        f32 = dict(device=dev, dtype=torch.float32)
        # static inputs, used when the caller's tensors are not taken in place (layout, alignment, too many address sets)
        self.V = torch.zeros((B, N, d), **f32)
        self.Q = [torch.zeros((B, T, d), **f32) for _ in range(3)]
        self.labels = torch.zeros((B,), device=dev, dtype=torch.int64)
        # static outputs
        self.v = torch.empty((3, B, d), **f32); self.q = torch.empty((3, B, d), **f32)
        # logits and loss share one buffer: a step hands out ONE copy of it (two views), not two
        self.out = torch.empty(B * K + 4, **f32)
        self.logits = self.out[:B * K].view(B, K); self.loss = self.out[B * K]
        self.dx = torch.empty((3, B, d), **f32)                      # d(q_l + v_l): upstream gradient of both v and q
        self.dV = torch.empty((B, N, d), **f32) if need_dv else None
        self.dQ = [torch.empty((B, T, d), **f32) for _ in range(3)]
        self.co_params, self.head_params = self._param_lists(co_attention, mlp_classify)
        for p in self.co_params + self.head_params:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise RuntimeError("HotPathGraph: parameters must be contiguous fp32")
        # (modules.HierarchicalCoAttentionNet checks them: the plans / graphs hold raw addresses of all 16 parameters, and
        #  .to(), p.data = ..., load_state_dict(assign=True) re-point storage without touching the Parameter objects)
        self.param_ptrs = tuple(p.data_ptr() for p in self.co_params + self.head_params)
        self.param_ptr0 = self.param_ptrs[0]
        self.co_grads = [torch.empty_like(p) for p in self.co_params]
        self.head_grads = [torch.empty_like(p) for p in self.head_params]
        sb, fb, bb = _lib.workspace_bytes(B, N, T, d, 3, flags)
        hsb, hwb = _head_ws(B, d, mlp, K)
        self.saved = torch.empty(sb // 4, **f32); self.ws = torch.empty(max(fb, bb) // 4, **f32)
        self.hsaved = torch.empty(hsb // 4, **f32); self.hws = torch.empty(hwb // 4, **f32)
        self.g_loss = torch.ones(1, **f32)                           # upstream gradient of the loss (device scalar)
        self._pairs = {}                                             # input addresses -> (forward graph, backward graph)
        self._plans = {}                                             # input addresses -> argument blocks of the four calls
        self._warm = False
        self._static = (self.V, self.Q[0], self.Q[1], self.Q[2], self.labels)
        self._lib = _lib.load()
        self.pair(self._static)

    @staticmethod
    def _param_lists(co, mlp):
        return ([co.W_v.weight, co.W_v.bias, co.W_q.weight, co.W_q.bias, co.w_v.weight, co.w_v.bias, co.w_q.weight, co.w_q.bias],
                [mlp.W_w.weight, mlp.W_w.bias, mlp.W_p.weight, mlp.W_p.bias, mlp.W_s.weight, mlp.W_s.bias, mlp.W_h.weight, mlp.W_h.bias])

    def stale(self) -> bool:
        """True if any of the 16 parameters this node reads by raw address is no longer the module's Parameter object, or
        has had its storage re-pointed (.to(), p.data = ..., load_state_dict(assign=True)): the node must be rebuilt."""
        co, head = self._param_lists(self.co, self.mlp)
        now = co + head
        mine = self.co_params + self.head_params
        return any(a is not b for a, b in zip(now, mine)) or tuple(p.data_ptr() for p in now) != self.param_ptrs

    def _key(self, ins):
        V = ins[0]
        return (V.data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), ins[3].data_ptr(), ins[4].data_ptr(), V.stride(0), V.stride(1), V.stride(2))

    def _plan(self, ins, key=None):
        """Argument blocks of the four C-ABI calls for the inputs `ins` = (V [B,N,d] in a native layout, Q_w, Q_p, Q_s,
        labels), built once per address set (everything but the stream).  Parameters are read where they lie (the
        optimiser updates them in place); the ctypes blocks are kept alive by the plan."""
        key = key or self._key(ins)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        if len(self._plans) >= 8 * self.MAX_KEYS:                # (eager mode has no graph to keep: drop the oldest)
            self._plans.pop(next(iter(self._plans)))
        B, N, T, d, mlp, K = self.dims
        V, Qs, labels = ins[0], ins[1:4], ins[4]
        qptr = (C.c_void_p * 3)(*[t.data_ptr() for t in Qs])
        dqptr = (C.c_void_p * 3)(*[t.data_ptr() for t in self.dQ])
        rows = lambda t: (C.c_void_p * 3)(*[t[l].data_ptr() for l in range(3)])   # noqa: E731
        p = _lib.Params(*[t.data_ptr() for t in self.co_params])
        pg = _lib.ParamGrads(*[t.data_ptr() for t in self.co_grads])
        hp = _lib.HeadParams(*[t.data_ptr() for t in self.head_params])
        hg = _lib.HeadParamGrads(*[t.data_ptr() for t in self.head_grads])
        rv, rq, rdx = rows(self.v), rows(self.q), rows(self.dx)
        vs = _strides(V)
        dvs = (N * d, d, 1) if self.dV is not None else (0, 0, 0)   # the static dV buffer is location-major
        plan = {
            "keep": (qptr, dqptr, p, pg, hp, hg, rv, rq, rdx),
            "co_fwd": (_ptr(V), *vs, qptr, C.byref(p), _ptr(self.v), _ptr(self.q), _ptr(self.saved), _ptr(self.ws),
                       B, N, T, d, 3, _lib.F32, self.flags),
            "head_fwd": (rv, rq, C.byref(hp), _ptr(labels), _ptr(self.logits), _ptr(self.loss), _ptr(self.hsaved),
                         B, d, mlp, K, _lib.F32, self.head_flags),
            "head_bwd": (rv, rq, C.byref(hp), _ptr(self.hsaved), _ptr(self.g_loss), None, rdx, None, C.byref(hg), 0,
                         _ptr(self.hws), B, d, mlp, K, _lib.F32, self.head_flags),
            "co_bwd": (_ptr(V), *vs, qptr, C.byref(p), _ptr(self.saved), _ptr(self.dx), _ptr(self.dx), _ptr(self.dV), *dvs,
                       dqptr, C.byref(pg), 0, _ptr(self.ws), B, N, T, d, 3, _lib.F32, self.flags),
        }
        # (variants of the two backward calls: a gradient arriving for the logits; accumulate = 1 for a second backward
        #  onto the same static gradient buffers; the loss's upstream gradient read where autograd left it)
        plan["head_bwd_args"] = lambda g_logits, acc, g_loss=None: (plan["head_bwd"][:4] + (_ptr(self.g_loss if g_loss is None else g_loss), _ptr(g_logits))   # noqa: E731
                                                                    + plan["head_bwd"][6:9] + (acc,) + plan["head_bwd"][10:])
        plan["co_bwd_args"] = lambda acc: plan["co_bwd"][:15] + (acc,) + plan["co_bwd"][16:]   # noqa: E731
        assert plan["head_bwd"][9] == 0 and plan["head_bwd"][5] is None and plan["co_bwd"][15] == 0
        self._plans[key] = plan
        return plan

    # the C-ABI calls on `stream`, reading the inputs `ins` = (V [B,N,d] in a native layout, Q_w, Q_p, Q_s, labels)
    def _enqueue(self, ins, stream, fwd=True, bwd=True):
        lib = self._lib
        plan = self._plan(ins)
        st = C.c_void_p(stream)
        if fwd:
            _lib.check(lib.coattn_forward(*plan["co_fwd"], st), "coattn_forward")
            _lib.check(lib.coattn_head_forward(*plan["head_fwd"], st), "coattn_head_forward")
        if bwd:
            _lib.check(lib.coattn_head_backward(*plan["head_bwd"], st), "coattn_head_backward")
            _lib.check(lib.coattn_backward(*plan["co_bwd"], st), "coattn_backward")

    def usable_in_place(self, ins) -> bool:
        B, N, T, d, _, _ = self.dims
        V, labels = ins[0], ins[4]
        # (labels: the kernels read `const long long*` at the captured address -- anything but contiguous int64 on this
        #  device goes through the static copy, which converts)
        return (V.dtype == torch.float32 and V.device == self.device and _native_layout(V) is V
                and labels.is_contiguous() and labels.dtype == torch.int64 and labels.device == self.device
                and all(q.is_contiguous() and q.dtype == torch.float32 and q.data_ptr() % 16 == 0 and q.device == self.device
                        for q in ins[1:4]))

    def pair(self, ins, key=None):
        """(forward graph, backward graph) reading the inputs at the addresses of `ins`; captured on first use, or
        None when MAX_KEYS address sets are already held (the caller then copies into the static inputs)."""
        if not self.capture:                                     # eager mode: nothing to capture (run() issues the calls)
            return _EAGER
        key = key or self._key(ins)
        hit = self._pairs.get(key)
        if hit is not None:
            return hit
        if len(self._pairs) >= self.MAX_KEYS:
            return None
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if not self._warm:                                   # per-device one-time setup must happen outside capture
                side = torch.cuda.Stream(self.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._enqueue(ins, side.cuda_stream)
                cur.wait_stream(side)
                self._warm = True
            torch.cuda.synchronize(self.device)
            gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread_local: other threads (train.DevicePrefetcher pins and copies batches concurrently) may keep calling
            # the allocator and the copy engine while this thread captures
            with torch.cuda.graph(gf, capture_error_mode="thread_local"):
                self._enqueue(ins, torch.cuda.current_stream(self.device).cuda_stream, True, False)
            with torch.cuda.graph(gb, pool=gf.pool(), capture_error_mode="thread_local"):
                self._enqueue(ins, torch.cuda.current_stream(self.device).cuda_stream, False, True)
        self._pairs[key] = (gf, gb)
        return self._pairs[key]

    def run_eager(self, ins=None):
        """The same calls without the graphs (tests compare the two bit for bit)."""
        self._enqueue(ins or self._static, torch.cuda.current_stream(self.device).cuda_stream)

    def run(self, pair, ins, fwd: bool, plan=None, g_logits=None, accumulate: int = 0, g_loss=None):
        """One direction of the hot path: replay the captured graph, or (eager mode) issue its two C-ABI calls (`plan`: the
        argument blocks of `ins`, if the caller has them already).  g_logits / accumulate (backward, eager mode only): an
        upstream gradient of the logits to add; add into the parameter-gradient buffers instead of overwriting them."""
        if pair is not _EAGER and (g_logits is not None or accumulate):
            raise RuntimeError("HotPathGraph: a gradient for `logits` / a second backward onto the same gradient buffers "
                               "cannot be replayed from the captured graph (use capture=False, or the module path)")
        if pair is _EAGER and not fwd and (g_logits is not None or accumulate or g_loss is not None):
            plan = plan or self._plan(ins)
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            with _lib.on_device(self.device):
                _lib.check(self._lib.coattn_head_backward(*plan["head_bwd_args"](g_logits, accumulate, g_loss), st), "coattn_head_backward")
                _lib.check(self._lib.coattn_backward(*plan["co_bwd_args"](accumulate), st), "coattn_backward")
            return
        if pair is _EAGER:
            plan = plan or self._plan(ins)
            lib = self._lib
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            with _lib.on_device(self.device):
                if fwd:
                    _lib.check(lib.coattn_forward(*plan["co_fwd"], st), "coattn_forward")
                    _lib.check(lib.coattn_head_forward(*plan["head_fwd"], st), "coattn_head_forward")
                else:
                    _lib.check(lib.coattn_head_backward(*plan["head_bwd"], st), "coattn_head_backward")
                    _lib.check(lib.coattn_backward(*plan["co_bwd"], st), "coattn_backward")
        else:
            pair[0 if fwd else 1].replay()

    def replay(self, ins=None):
        ins = ins or self._static
        pair = self.pair(ins)
        self.run(pair, ins, True)
        self.run(pair, ins, False)

    def __call__(self, x_img: torch.Tensor, x_ques: Sequence[torch.Tensor], labels: torch.Tensor):
        # With direct gradients the parameters need not be inputs of the autograd node (their gradients do not travel through
        # autograd) as long as some input keeps the node alive -- the question levels of a trainable question encoder do;
        # 6 arguments instead of 22 through the Function machinery on every step.
        # Features of a frozen encoder that the kernels do not take where they lie (bf16 from an autocast encoder; the
        # channel-major view at N = 49) go through the library's one-pass conversion into the node's static fp32 buffer --
        # one address set for the plans / graphs, and neither autocast's up-cast nor torch's strided copy on top of it.
        if (x_img.is_cuda and not (x_img.requires_grad and torch.is_grad_enabled()) and x_img.data_ptr() != self.V.data_ptr()
                and x_img.dtype in (torch.float32, torch.bfloat16) and tuple(x_img.shape) == tuple(self.V.shape)
                and x_img.device == self.device
                and (x_img.dtype != torch.float32 or not _is_native(x_img, lm_only=_coattention.CM_FEATURES != "inplace"))):
            x_img = native_features(x_img, out=self.V)
        if self.direct_grads and (x_ques[0].requires_grad or x_ques[1].requires_grad or x_ques[2].requires_grad or x_img.requires_grad):
            return _HotPathFn.apply(self, x_img, labels, *x_ques)
        return _HotPathFn.apply(self, x_img, labels, *x_ques, *self.co_params, *self.head_params)


_EAGER = ("eager", "eager")        # what pair() returns in eager mode (capture=False)


class _HotPathFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, hp: HotPathGraph, x_img, labels, Qw, Qp, Qs, *params):
        B, N, T, d, mlp, K = hp.dims
        if tuple(x_img.shape) != (B, N, d) or any(tuple(q.shape) != (B, T, d) for q in (Qw, Qp, Qs)) or tuple(labels.shape) != (B,):
            raise RuntimeError("HotPathGraph: captured for x_img %s, questions %s" % ((B, N, d), (B, T, d)))
        if labels.dtype.is_floating_point or labels.dtype == torch.bool:
            raise RuntimeError("HotPathGraph: labels must be integer class indices (int64 [B]), got %s" % labels.dtype)
        if ctx.needs_input_grad[1] and hp.dV is None:
            raise RuntimeError("HotPathGraph: built with need_dv=False but the image features require a gradient")
        # (features in a layout the kernels do not run on -- e.g. the channel-major view at N = 49, whose rows are not
        #  16-byte multiples -- are re-laid once, as on the eager path; the allocator hands that buffer's block back step
        #  after step, so it is one more address set, not a copy into the static input on top)
        ins = (_native_layout(x_img), Qw, Qp, Qs, labels)
        key = hp._key(ins)
        pair = hp.pair(ins, key) if hp.usable_in_place(ins) else None
        if pair is None:                                         # other dtype / too many address sets: static inputs
            if x_img.data_ptr() != hp.V.data_ptr():
                hp.V.copy_(x_img)
            torch._foreach_copy_(hp.Q, [Qw, Qp, Qs])
            hp.labels.copy_(labels)
            ins = hp._static
            key = hp._key(ins)
            pair = hp.pair(ins, key)
        plan = hp._plan(ins, key) if pair is _EAGER else None   # (built once per address set; the backward reuses it)
        hp.run(pair, ins, True, plan)
        ctx.plan = plan
        # (as head.answer_head after a forward with labels: Trainer.check_labels() reads this step's status word)
        _head._last = (hp.hsaved, B, d, mlp, K, hp.device)
        if hp.flags & _lib.FLAG_FAST16:                          # tolerance mode: this step's status words (_lib.check_range())
            _lib.note_status("coattn", hp.saved, (B, N, T, d, 3), hp.device)
        ctx.hp, ctx.pair = hp, pair
        ctx.keep = ins                                           # the graphs read these addresses again in backward
        ctx.set_materialize_grads(False)                         # (an unused output arrives as None, not as zeros)
        # logits: a fresh tensor by default -- the static buffer is overwritten by the next step
        if hp.alias_outputs:
            return hp.logits, hp.loss.clone()
        out = hp.out.clone()                                      # (one copy kernel for both)
        return out[:B * K].view(B, K), out[B * K]

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_logits, g_loss):
        hp = ctx.hp
        if g_loss is None and g_logits is None:
            return (None,) * len(ctx.needs_input_grad)
        direct = None
        if g_loss is None:
            hp.g_loss.zero_()
        elif (ctx.pair is _EAGER and g_loss.dtype == torch.float32 and g_loss.device == hp.device and g_loss.numel() == 1
              and g_loss.data_ptr() % 4 == 0):
            direct = g_loss                                      # eager calls read it where autograd left it: no copy kernel
            # (the kernel reads it after this function has returned and autograd has dropped it: tell the allocator which
            #  stream still uses the block, in case it was allocated on another one)
            g_loss.record_stream(torch.cuda.current_stream(hp.device))
        else:
            hp.g_loss.copy_(g_loss.reshape(1))                   # (the captured graph reads the static scalar)
        if g_logits is not None:
            g_logits = g_logits.contiguous().float()
        acc = 0
        if hp.direct_grads:
            # A second backward with the static buffers still in place as param.grad (no zero_grad(set_to_none=True) in
            # between: micro-batch accumulation, retain_graph): the kernels must ADD, or the earlier gradient is lost
            held = [p.grad is g for p, g in zip(hp.co_params + hp.head_params, hp.co_grads + hp.head_grads)]
            if all(held):
                acc = 1
                if not hp._warned_accumulate:
                    # (inferred from identity: a loop that never drops its gradients -- zero_grad(set_to_none=False), or no
                    #  zero_grad at all -- lands here too, and an in-place clip / scale of p.grad has changed what is added to)
                    hp._warned_accumulate = True
                    import warnings
                    warnings.warn("HotPathGraph(direct_grads=True): param.grad still holds the static gradient buffers of the "
                                  "previous backward, so this backward ADDS into them (micro-batch accumulation).  If the "
                                  "gradients were meant to be overwritten, call zero_grad(set_to_none=True) before every backward.")
            elif any(held):
                raise RuntimeError("HotPathGraph(direct_grads=True): some parameters still hold the static gradient buffer "
                                   "from the previous backward and others do not -- call zero_grad(set_to_none=True) on all of "
                                   "them before every backward, or on none")
        hp.run(ctx.pair, ctx.keep, False, ctx.plan, g_logits=g_logits, accumulate=acc, g_loss=direct)
        if hp.direct_grads:
            # the owner of the step (train.Trainer) consumes the gradients before the next backward and needs no gradient
            # hooks: the static buffers BECOME param.grad (autograd's AccumulateGrad would clone each one -- it cannot
            # steal a buffer this object still holds: 16 copy kernels per step)
            for p, g in zip(hp.co_params + hp.head_params, hp.co_grads + hp.head_grads):
                if p.grad is None or p.grad is g:
                    p.grad = g
                else:
                    p.grad.add_(g)
            return (None, hp.dV, None, *hp.dQ) + (None,) * (len(ctx.needs_input_grad) - 6)
        # (the static gradient buffers are handed out as they are: autograd accumulates / the optimiser consumes them
        #  before the next step's backward overwrites them)
        return (None, hp.dV, None, *hp.dQ, *hp.co_grads, *hp.head_grads)
