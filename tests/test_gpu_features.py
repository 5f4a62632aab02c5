"""coattn_features_native (include/coattn.h): image features as the encoder leaves them -> fp32 location-major, through
the C-ABI; and its callers (vqa_amd.native_features, the module, the hot-path node).  The reference side of this boundary is
model.py:215-217 (view + permute of the NCHW conv features) under main.py:73, :185 (AMP activations)."""
import ctypes as C

import pytest
import torch

import vqa_amd
from vqa_amd import _lib
from vqa_amd.coattention import native_features

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call(x, out):
    lib = _lib.load()
    B, N, d = x.shape
    sB, sN, sD = x.stride()
    rc = lib.coattn_features_native(C.c_void_p(x.data_ptr()), _lib.BF16 if x.dtype == torch.bfloat16 else _lib.F32,
                                    sB, sN, sD, C.c_void_p(out.data_ptr()), B, N, d,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "coattn_features_native")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,d", [(3, 49, 512), (2, 196, 512), (2, 49, 2048), (1, 1, 64), (2, 50, 96), (3, 7, 130), (1, 65, 63)])
def test_channel_major_view_to_native(dtype, B, N, d):
    """the permuted NCHW view, rows of N elements (unaligned at N = 49): bit-exact against torch's up-cast + copy"""
    torch.manual_seed(0)
    x = torch.randn(B, d, N, device=DEV).to(dtype).permute(0, 2, 1)      # [B,N,d], strides (d N, 1, N)
    out = torch.full((B, N, d), float("nan"), device=DEV)
    _call(x, out)
    assert torch.equal(out, x.float().contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_other_strides(dtype):
    """location-major with a padded sample stride, a sliced channel range, a broadcast sample"""
    torch.manual_seed(1)
    base = torch.randn(3, 40, 200, device=DEV).to(dtype)
    for x in (base[:, :33, 8:136], base[:, ::2, :128], base[:1].expand(3, 40, 200), base.permute(0, 2, 1)[:, 3:150, :]):
        out = torch.full(tuple(x.shape), float("nan"), device=DEV)
        _call(x, out)
        assert torch.equal(out, x.float().contiguous())


def test_bad_arguments_are_refused():
    lib = _lib.load()
    x = torch.zeros(2, 4, 8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.coattn_features_native(C.c_void_p(x.data_ptr()), 7, 32, 8, 1, C.c_void_p(x.data_ptr()), 2, 4, 8, st) != 0
    assert b"dtype" in lib.coattn_last_error()
    assert lib.coattn_features_native(None, _lib.F32, 32, 8, 1, C.c_void_p(x.data_ptr()), 2, 4, 8, st) != 0
    assert lib.coattn_features_native(C.c_void_p(x.data_ptr()), _lib.F32, 32, -8, 1, C.c_void_p(x.data_ptr()), 2, 4, 8, st) != 0


def test_native_features_helper():
    torch.manual_seed(2)
    lm = torch.randn(2, 49, 512, device=DEV)
    assert native_features(lm) is lm                                      # fp32 location-major: where it lies
    cm196 = torch.randn(2, 512, 196, device=DEV).permute(0, 2, 1)
    y = native_features(cm196)                                            # frozen channel-major features: converted (faster kernels)
    assert y.is_contiguous() and torch.equal(y, cm196.contiguous())
    import sys
    ca = sys.modules["vqa_amd.coattention"]                               # (the package attribute of that name is the function)
    old, ca.CM_FEATURES = ca.CM_FEATURES, "inplace"
    try:
        assert native_features(cm196) is cm196                            # ... or read where they lie (16-byte rows)
    finally:
        ca.CM_FEATURES = old
    cm49 = torch.randn(2, 512, 49, device=DEV).permute(0, 2, 1)
    y = native_features(cm49)
    assert y.is_contiguous() and torch.equal(y, cm49.contiguous())
    bf = cm49.to(torch.bfloat16)
    buf = torch.empty(2, 49, 512, device=DEV)
    y = native_features(bf, out=buf)
    assert y is buf and torch.equal(y, bf.float().contiguous())
    with pytest.raises(RuntimeError, match="gradient"):
        native_features(lm.clone().requires_grad_(True))


def test_module_takes_bf16_channel_major_features_of_a_frozen_encoder():
    """the module fed what an autocast NCHW encoder leaves (bf16, permuted view at N = 49) equals the module fed the same
    values as fp32 location-major -- forward outputs and question-side / parameter gradients, bit for bit"""
    torch.manual_seed(3)
    B, N, T, d = 4, 49, 26, 512
    co = vqa_amd.ParallelCoAttention(d).to(DEV)
    feats = torch.randn(B, d, N, device=DEV).clamp_min_(0).to(torch.bfloat16).permute(0, 2, 1)
    ref_in = feats.float().contiguous()
    res = []
    for x in (feats, ref_in):
        co.zero_grad(set_to_none=True)
        Qs = [torch.randn(B, T, d, device=DEV, generator=torch.Generator(DEV).manual_seed(5 + l)).requires_grad_(True) for l in range(3)]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=x.dtype == torch.bfloat16):
            vs, qs = co(x, Qs)
        loss = sum((v * v).sum() for v in vs) + sum((q * q).sum() for q in qs)
        loss.backward()
        res.append(([v.detach().clone() for v in vs + qs], [q.grad.clone() for q in Qs], co.W_v.weight.grad.clone()))
    for a, b in zip(res[0][0] + res[0][1] + [res[0][2]], res[1][0] + res[1][1] + [res[1][2]]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("capture", [False, True])
def test_hot_path_node_converts_into_its_static_buffer(capture):
    """HotPathGraph fed bf16 channel-major features: one library pass into its static V, same loss / logits / gradients as
    the node fed the fp32 location-major values"""
    from vqa_amd.graph import HotPathGraph
    from vqa_amd.modules import MLPClassifier
    torch.manual_seed(4)
    B, N, T, d, K = 8, 49, 26, 512, 101
    co = vqa_amd.ParallelCoAttention(d).to(DEV)
    head = MLPClassifier(d, 256, K).to(DEV)
    feats = torch.randn(B, d, N, device=DEV).clamp_min_(0).to(torch.bfloat16).permute(0, 2, 1)
    labels = torch.randint(0, K, (B,), device=DEV)
    out = []
    for x in (feats, feats.float().contiguous()):
        hp = HotPathGraph(co, head, B, N, T, need_dv=False, capture=capture)
        for p in list(co.parameters()) + list(head.parameters()):
            p.grad = None
        Qs = [torch.randn(B, T, d, device=DEV, generator=torch.Generator(DEV).manual_seed(9 + l)).requires_grad_(True) for l in range(3)]
        logits, loss = hp(x, Qs, labels)
        loss.backward()
        torch.cuda.synchronize()
        if x.dtype == torch.bfloat16:
            assert torch.equal(hp.V, x.float().contiguous())
        out.append((logits.clone(), loss.clone(), [q.grad.clone() for q in Qs], co.W_q.weight.grad.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2] + [out[0][3]], out[1][2] + [out[1][3]]):
        assert torch.equal(a, b)
