"""ctypes binding of ``libcoattn_hip.so`` (C-ABI declared in ``include/coattn.h``)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# COATTN_LIB_PATH: developer override for A/B timing of two builds; the product loads the in-tree library
LIB_PATH = os.environ.get("COATTN_LIB_PATH") or os.path.join(_HERE, "libcoattn_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/coattn.h declares
EXPORTS = ("coattn_version", "coattn_last_error", "coattn_fused_supported", "coattn_workspace_bytes",
           "coattn_forward", "coattn_attention_forward", "coattn_backward", "coattn_gemm_f32", "coattn_gemm_bf16",
           "coattn_phrase_workspace_bytes", "coattn_phrase_forward", "coattn_phrase_backward",
           "coattn_ce_workspace_bytes", "coattn_ce_forward", "coattn_linear_workspace_bytes", "coattn_linear_forward",
           "coattn_linear_wgrad_workspace_bytes", "coattn_linear_weight_grad",
           "coattn_head_workspace_bytes", "coattn_head_forward", "coattn_head_backward", "coattn_head_status",
           "coattn_ce_status", "coattn_p2p_enable_peer", "coattn_p2p_reduce_scatter", "coattn_p2p_all_gather",
           "coattn_profile_begin", "coattn_profile_end", "coattn_features_native", "coattn_status",
           "coattn_phrase_status", "coattn_status_accumulate", "coattn_phrase_status_accumulate")

F32 = 0
BF16 = 1                  # storage type of coattn_features_native's input
IMPL_AUTO, IMPL_GENERAL, IMPL_FUSED = 0, 1, 2
FLAG_BF16_PROJ = 4
FLAG_BF16_IN = 8          # linear entry points: x (dy) stored as bf16
FLAG_EXACT3 = 16          # coattn_forward / coattn_backward: every contraction on the exact three-piece split (= flags 0: the default)
FLAG_SPLIT2 = 32          # linear entry points: the two-piece width (hi + mid, three partial products)
FLAG_F16PAIR = 64        # coattn_linear_forward: two FP16 pieces (the form the tolerance mode runs its projections in)
FLAG_FAST16 = 128        # coattn_forward / coattn_backward / coattn_phrase_*: the tolerance mode (forward products on two FP16
                         # pieces = 22 bits, backward on two bf16 pieces = 16 bits; range report: coattn_status)


def precision_flag(fast: bool) -> int:
    return FLAG_FAST16 if fast else 0


def default_fast() -> bool:
    """The modules' default precision: exact fp32 products (the reference's arithmetic) unless VQA_PRECISION=fast;
    train.Trainer opts into the tolerance mode itself (precision="fast")."""
    return os.environ.get("VQA_PRECISION", "exact") == "fast"


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W_v", "b_v", "W_q", "b_q", "w_v", "c_v", "w_q", "c_q")]


class ParamGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dW_v", "db_v", "dW_q", "db_q", "dw_v", "dc_v", "dw_q", "dc_q")]


class PhraseParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3")]


class PhraseParamGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dW1", "db1", "dW2", "db2", "dW3", "db3")]


class HeadParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W_w", "b_w", "W_p", "b_p", "W_s", "b_s", "W_h", "b_h")]


class HeadParamGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dW_w", "db_w", "dW_p", "db_p", "dW_s", "db_s", "dW_h", "db_h")]


class GemmDesc(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in ("A", "B", "Cin", "C", "bias_n", "bias_m")]
                + [(n, C.c_int) for n in ("M", "N", "K", "batch", "inner", "inner_total", "ksplit", "act")]
                + [("beta", C.c_float)]
                + [(n, C.c_int64) for n in ("a_sm", "a_sk", "a_sz", "a_si", "a_mdiv", "a_sdiv",
                                            "b_sk", "b_sn", "b_sz", "b_si",
                                            "c_sm", "c_sn", "c_sz", "c_mdiv", "c_sdiv",
                                            "cin_sm", "cin_sn", "cin_sz", "cin_mdiv", "cin_sdiv")]
                + [("a_ptrs", C.c_void_p * 8), ("b_ptrs", C.c_void_p * 8), ("c_ptrs", C.c_void_p * 8),
                   ("cin_ptrs", C.c_void_p * 8), ("ptr_by_inner", C.c_int), ("b_imod", C.c_int),
                   ("kband_n", C.c_int), ("kband_lo", C.c_int * 3), ("kband_hi", C.c_int * 3),
                   ("out_scale", C.c_float)])


def build(verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j", "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libcoattn_hip.so failed")
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    """Load the library (after torch, so that the HIP runtime torch ships is the one bound)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- must be imported first: one HIP runtime per process

    if not os.path.isfile(LIB_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the co-attention path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError("libcoattn_hip.so lacks symbol %s" % name)
    lib.coattn_version.restype = C.c_int
    lib.coattn_last_error.restype = C.c_char_p
    lib.coattn_fused_supported.argtypes = [C.c_int] * 6
    lib.coattn_workspace_bytes.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_size_t)] * 3
    lib.coattn_forward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(Params),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    lib.coattn_attention_forward.argtypes = lib.coattn_forward.argtypes
    lib.coattn_backward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(Params),
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                    C.POINTER(C.c_void_p), C.POINTER(ParamGrads), C.c_int,
                                    C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
    lib.coattn_gemm_f32.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    lib.coattn_gemm_bf16.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    lib.coattn_phrase_workspace_bytes.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_size_t)] * 3
    lib.coattn_phrase_forward.argtypes = [C.c_void_p, C.POINTER(PhraseParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.coattn_phrase_backward.argtypes = [C.c_void_p, C.POINTER(PhraseParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(PhraseParamGrads), C.c_int, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.coattn_ce_workspace_bytes.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_size_t)]
    lib.coattn_ce_forward.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]
    lib.coattn_linear_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.coattn_linear_workspace_bytes.restype = C.c_size_t
    lib.coattn_linear_forward.argtypes = ([C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int] * 3
                                          + [C.c_float, C.c_int, C.c_void_p])
    lib.coattn_linear_wgrad_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.coattn_linear_wgrad_workspace_bytes.restype = C.c_size_t
    lib.coattn_linear_weight_grad.argtypes = ([C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
                                              + [C.c_int] * 4 + [C.c_void_p])
    lib.coattn_ce_status.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.coattn_status.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.POINTER(C.c_float)]
    lib.coattn_phrase_status.argtypes = [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.POINTER(C.c_float)]
    lib.coattn_status_accumulate.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
    lib.coattn_phrase_status_accumulate.argtypes = [C.c_void_p] + [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
    lib.coattn_p2p_enable_peer.argtypes = [C.c_int]
    lib.coattn_p2p_reduce_scatter.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.c_float, C.c_void_p]
    lib.coattn_p2p_all_gather.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.c_void_p]
    lib.coattn_profile_begin.argtypes = [C.c_void_p]
    lib.coattn_profile_end.argtypes = [C.POINTER(C.c_float), C.c_char_p, C.c_int, C.c_int]
    lib.coattn_features_native.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.coattn_head_status.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    lib.coattn_head_workspace_bytes.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_size_t)] * 2
    lib.coattn_head_forward.argtypes = ([C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(HeadParams)]
                                        + [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p])
    lib.coattn_head_backward.argtypes = ([C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(HeadParams)]
                                         + [C.c_void_p] * 3 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                               C.POINTER(HeadParamGrads), C.c_int, C.c_void_p]
                                         + [C.c_int] * 6 + [C.c_void_p])
    _lib = lib
    return lib


class RangeError(FloatingPointError):
    """An operand of the tolerance mode (FLAG_FAST16) left the exact range of its two FP16 pieces: the call's results are
    clamped, not within tolerance of the reference.  Use the exact mode (the modules' default; Trainer(precision="exact"))."""


F16_EXACT = 65504.0

# Range report of the tolerance mode, sticky: every tolerance-mode forward notes where its status words are (note_status),
# and the note is folded -- asynchronously, one tiny launch on the stream the forward ran on -- into a two-float accumulator
# per device (include/coattn.h coattn_status_accumulate); check_range() reads the accumulators back (synchronising), clears
# them and raises if ANY forward since the last check left the FP16-piece range.  A later forward, a validate() pass or a
# second model cannot overwrite an earlier call's report: the accumulator only grows until it is read.
_pending = []                 # (kind, tensor holding the status words, dims, device, stream pointer)
_range_acc = {}               # device index -> float32[2] accumulator


def note_status(kind: str, buf, dims, dev) -> None:
    import torch
    _pending.append((kind, buf, dims, dev, torch.cuda.current_stream(dev).cuda_stream))
    # folded right away, behind the forward on its own stream: the holder may be a static buffer (graph.py) or the
    # per-stream scratch (forward without `saved`), which the next call rewrites -- except while the stream is being
    # captured into a graph (the note then waits for the first fold outside the capture)
    if not torch.cuda.is_current_stream_capturing():
        fold_range()


def _acc(dev):
    import torch
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    a = _range_acc.get(key)
    if a is None:
        a = _range_acc[key] = torch.zeros(2, device=dev, dtype=torch.float32)
    return a


def fold_range() -> None:
    """Fold the status words of every tolerance-mode forward noted since the last fold into the per-device accumulators.
    Asynchronous (one 256-thread launch per call, on the stream the call ran on)."""
    lib = load()
    while _pending:
        kind, buf, dims, dev, st = _pending.pop(0)
        acc = _acc(dev)
        with on_device(dev):
            if kind == "coattn":
                rc = lib.coattn_status_accumulate(C.c_void_p(buf.data_ptr()), *dims, F32, C.c_void_p(acc.data_ptr()), C.c_void_p(st))
            else:
                rc = lib.coattn_phrase_status_accumulate(C.c_void_p(buf.data_ptr()), *dims, C.c_void_p(acc.data_ptr()), C.c_void_p(st))
        check(rc, "coattn_status_accumulate" if kind == "coattn" else "coattn_phrase_status_accumulate")


def range_maxima(reset: bool = True):
    """(largest out-of-range |activation|, largest |256 W|) over every tolerance-mode forward since the last reset, over
    all devices of this process.  Synchronises."""
    fold_range()
    act = wgt = 0.0
    for a in _range_acc.values():
        h = a.cpu()                                      # (synchronises the device)
        x, w = float(h[0]), float(h[1])
        act = x if (x > act or x != x) else act
        wgt = w if (w > wgt or w != w) else wgt
        if reset:
            a.zero_()
    return act, wgt


def check_range(stream=None) -> None:
    """Raise RangeError if ANY tolerance-mode co-attention / phrase forward since the last check met an operand outside the
    FP16-piece range (include/coattn.h, coattn_status_accumulate).  Synchronises -- call it where the host reads the loss.
    (`stream` is accepted for compatibility with v0.6.0 callers and ignored: every note is folded on the stream of its own
    forward.)"""
    del stream
    act, wgt = range_maxima()
    bad_a, bad_w = not (act <= F16_EXACT), not (wgt <= F16_EXACT)
    if bad_a or bad_w:
        raise RangeError("FP16-piece range exceeded in a tolerance-mode forward (COATTN_FLAG_FAST16): %s%s%s -- pieces were "
                         "clamped; use the exact mode (largest activation beyond the range %.4g, largest 256*|W| %.4g)"
                         % ("an activation (feature or stored projection) of magnitude > 65504" if bad_a else "",
                            " and " if bad_a and bad_w else "",
                            "a projection weight of magnitude > 255.87" if bad_w else "", act, wgt))


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().coattn_last_error().decode()))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """Context manager that makes `dev` the current device for the C-ABI call inside (per-device one-time kernel
    attributes are keyed by it) -- a no-op object when it already is: ``torch.cuda.device`` costs ~10 us of host time per
    entry, the hot path enters it four times per step."""
    import torch
    return _NO_GUARD if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


_ws_cache = {}


def workspace_bytes(B, N, T, d, L, flags=0):
    """(saved, ws_fwd, ws_bwd) in bytes; cached per shape (the plan is a pure function of the shape)."""
    key = (B, N, T, d, L, flags)
    hit = _ws_cache.get(key)
    if hit is not None:
        return hit
    s, f, b = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(load().coattn_workspace_bytes(B, N, T, d, L, F32, flags, C.byref(s), C.byref(f), C.byref(b)),
          "coattn_workspace_bytes")
    _ws_cache[key] = (s.value, f.value, b.value)
    return _ws_cache[key]


_scratch = {}
_SCRATCH_MAX_STREAMS = 8


def scratch(nbytes: int, device, stream_ptr: int):
    """Scratch workspace (contents undefined after a call) kept per (device, stream): calls on one stream run in
    order, so they can share it; grown on demand.  At most _SCRATCH_MAX_STREAMS buffers are kept (least recently used
    dropped: a process that touches many short-lived streams does not pile up workspaces), and nothing is cached while
    the stream is capturing a graph (the allocation then belongs to the graph's private pool)."""
    import torch
    if torch.cuda.is_current_stream_capturing():
        return torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream_ptr)
    buf = _scratch.pop(key, None)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
    _scratch[key] = buf                                  # (re-inserted last: dict order = recency)
    while len(_scratch) > _SCRATCH_MAX_STREAMS:
        _scratch.pop(next(iter(_scratch)))
    return buf
