"""GPU: the strided/batched MFMA GEMM building block (coattn_gemm_f32) against torch fp64."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _gemm(desc_kw, stream=None, bf16=False):
    from vqa_amd import _lib
    lib = _lib.load()
    g = _lib.GemmDesc()
    for k, v in desc_kw.items():
        setattr(g, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    keep = list(desc_kw.values())  # noqa: F841  (tensors / tables stay alive across the launch)
    fn = lib.coattn_gemm_bf16 if bf16 else lib.coattn_gemm_f32
    _lib.check(fn(C.byref(g), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    torch.cuda.synchronize()


def _r(x, bf16):
    """Reference operand: what the kernel multiplies (bf16 mode rounds to nearest even; products of two
    bf16 values are exact in fp32, so only the accumulation order differs from the float64 reference)."""
    return x.bfloat16().double() if bf16 else x.double()


def _rel(x, ref):
    return ((x.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item()


@pytest.mark.parametrize("M,N,K", [(26, 196, 512), (196, 512, 26), (130, 200, 70), (300, 96, 33), (5, 37, 96)])
def test_gemm_plain_rowmajor(M, N, K):
    torch.manual_seed(1)
    A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda")
    Cm = torch.full((M, N), float("nan"), device="cuda")
    _gemm(dict(A=A, B=B, C=Cm, M=M, N=N, K=K, batch=1, a_sm=K, a_sk=1, b_sk=N, b_sn=1, c_sm=N, c_sn=1))
    assert _rel(Cm, A.double() @ B.double()) < 2e-6


def test_gemm_transposed_operands_bias_beta_tanh():
    torch.manual_seed(2)
    M, N, K, Z = 70, 150, 45, 3
    At = torch.randn(Z, K, M, device="cuda") * 0.2      # A(m,k) = At[z][k][m]
    Bt = torch.randn(Z, N, K, device="cuda") * 0.2      # B(k,n) = Bt[z][n][k]
    bn = torch.randn(N, device="cuda"); bm = torch.randn(M, device="cuda")
    Cin = torch.randn(Z, M, N, device="cuda")
    Cm = torch.full((Z, M, N), float("nan"), device="cuda")
    _gemm(dict(A=At, B=Bt, Cin=Cin, C=Cm, bias_n=bn, bias_m=bm, M=M, N=N, K=K, batch=Z, act=1, beta=0.5,
               a_sm=1, a_sk=M, a_sz=K * M, b_sk=1, b_sn=K, b_sz=N * K,
               c_sm=N, c_sn=1, c_sz=M * N, cin_sm=N, cin_sn=1, cin_sz=M * N))
    ref = torch.tanh(At.double().transpose(1, 2) @ Bt.double().transpose(1, 2) + bn.double() + bm.double()[:, None]
                     + 0.5 * Cin.double())
    assert (Cm.double() - ref).abs().max().item() < 2e-6


def test_gemm_row_split_and_transposed_output():
    """A rows addressed as (m / N) * d*N + (m % N): the channel-major V of model.py:215-217."""
    torch.manual_seed(3)
    B_, d, N = 3, 96, 37
    V = torch.randn(B_, d, N, device="cuda"); W = torch.randn(d, d, device="cuda") / 10
    bias = torch.randn(d, device="cuda")
    Pv = torch.full((B_ * N, d), float("nan"), device="cuda")
    _gemm(dict(A=V, B=W, C=Pv, bias_n=bias, M=B_ * N, N=d, K=d, batch=1, a_sm=1, a_sk=N, a_mdiv=N, a_sdiv=d * N,
               b_sk=1, b_sn=d, c_sm=d, c_sn=1))
    ref = V.double().permute(0, 2, 1).reshape(B_ * N, d) @ W.double().T + bias.double()
    assert _rel(Pv, ref) < 2e-6
    # output rows split the same way (writes a [B,d,N] buffer from a [B*N, d] product)
    out = torch.full((B_, d, N), float("nan"), device="cuda")
    _gemm(dict(A=Pv, B=W, C=out, M=B_ * N, N=d, K=d, batch=1, a_sm=d, a_sk=1, b_sk=d, b_sn=1,
               c_sm=1, c_sn=N, c_mdiv=N, c_sdiv=d * N))
    ref2 = (Pv.double() @ W.double()).reshape(B_, N, d).permute(0, 2, 1)
    assert _rel(out, ref2) < 2e-6


def test_gemm_inner_groups_and_ksplit():
    torch.manual_seed(4)
    B_, N, d = 7, 37, 64
    dP = torch.randn(B_, N, d, device="cuda"); V = torch.randn(B_, d, N, device="cuda")
    G = 3; S = (B_ + G - 1) // G
    part = torch.full((S, d, d), float("nan"), device="cuda")
    _gemm(dict(A=dP, B=V, C=part, M=d, N=d, K=N, batch=S, inner=G, inner_total=B_,
               a_sm=1, a_sk=d, a_si=N * d, a_sz=G * N * d, b_sk=1, b_sn=N, b_si=d * N, b_sz=G * d * N,
               c_sm=d, c_sn=1, c_sz=d * d))
    ref = torch.einsum("bnj,bkn->jk", dP.double(), V.double())
    assert _rel(part.sum(0), ref) < 2e-6
    K = 200
    A = torch.randn(K, d, device="cuda"); Bm = torch.randn(K, d, device="cuda")
    ks = 48; S = (K + ks - 1) // ks
    part = torch.full((S, d, d), float("nan"), device="cuda")
    _gemm(dict(A=A, B=Bm, C=part, M=d, N=d, K=K, batch=S, ksplit=ks, a_sm=1, a_sk=d, b_sk=d, b_sn=1,
               c_sm=d, c_sn=1, c_sz=d * d))
    assert _rel(part.sum(0), A.double().T @ Bm.double()) < 2e-6


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("a_m,b_n", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_aligned_fast_path_layouts(a_m, b_n, bf16):
    """float4 / BK=32 kernel: all four operand layouts, partial edge tiles, K not a multiple of 32."""
    torch.manual_seed(5)
    M, N, K = 260, 200, 100
    A = torch.randn(K, M, device="cuda") if a_m else torch.randn(M, K, device="cuda")
    Bm = torch.randn(K, N, device="cuda") if b_n else torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    Cm = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(A=A, B=Bm, C=Cm, bias_n=bias, M=M, N=N, K=K, batch=1, c_sm=N, c_sn=1)
    kw.update(dict(a_sm=1, a_sk=M) if a_m else dict(a_sm=K, a_sk=1))
    kw.update(dict(b_sk=N, b_sn=1) if b_n else dict(b_sk=1, b_sn=K))
    _gemm(kw, bf16=bf16)
    ref = (_r(A, bf16).T if a_m else _r(A, bf16)) @ (_r(Bm, bf16) if b_n else _r(Bm, bf16).T) + bias.double()
    assert _rel(Cm, ref) < 2e-6


@pytest.mark.parametrize("bf16", [False, True])
def test_gemm_aligned_row_split_inner_ksplit_and_pointer_tables(bf16):
    torch.manual_seed(6)
    B_, d, N = 5, 128, 36
    V = torch.randn(B_, d, N, device="cuda"); W = torch.randn(d, d, device="cuda") / 10
    Pv = torch.full((B_ * N, d), float("nan"), device="cuda")
    _gemm(dict(A=V, B=W, C=Pv, M=B_ * N, N=d, K=d, batch=1, a_sm=1, a_sk=N, a_mdiv=N, a_sdiv=d * N,
               b_sk=1, b_sn=d, c_sm=d, c_sn=1), bf16=bf16)
    assert _rel(Pv, _r(V, bf16).permute(0, 2, 1).reshape(B_ * N, d) @ _r(W, bf16).T) < 2e-6
    # weight-gradient form: inner groups over samples
    dP = torch.randn(B_, N, d, device="cuda")
    G = 2; S = (B_ + G - 1) // G
    part = torch.full((S, d, d), float("nan"), device="cuda")
    _gemm(dict(A=dP, B=V, C=part, M=d, N=d, K=N, batch=S, inner=G, inner_total=B_,
               a_sm=1, a_sk=d, a_si=N * d, a_sz=G * N * d, b_sk=1, b_sn=N, b_si=d * N, b_sz=G * d * N,
               c_sm=d, c_sn=1, c_sz=d * d), bf16=bf16)
    assert _rel(part.sum(0), torch.einsum("bnj,bkn->jk", _r(dP, bf16), _r(V, bf16))) < 2e-6
    # level-merged launches: A / C through pointer tables, and B through a table indexed by the inner loop
    L, M = 3, 132
    Qs = [torch.randn(M, d, device="cuda") for _ in range(L)]
    out = torch.full((L, M, d), float("nan"), device="cuda")
    from vqa_amd import _lib
    import ctypes as C
    tab = (C.c_void_p * 8)(*([q.data_ptr() for q in Qs] + [None] * 5))
    _gemm(dict(a_ptrs=tab, B=W, C=out, M=M, N=d, K=d, batch=L, a_sm=d, a_sk=1, b_sk=1, b_sn=d,
               c_sm=d, c_sn=1, c_sz=M * d), bf16=bf16)
    for l in range(L):
        assert _rel(out[l], _r(Qs[l], bf16) @ _r(W, bf16).T) < 2e-6
    dPq = torch.randn(L, M, d, device="cuda")
    ks = 48; S = (M + ks - 1) // ks
    part = torch.full((S, d, d), float("nan"), device="cuda")
    _gemm(dict(A=dPq, b_ptrs=tab, ptr_by_inner=1, C=part, M=d, N=d, K=M, batch=S, ksplit=ks, inner=L,
               a_sm=1, a_sk=d, a_si=M * d, b_sk=d, b_sn=1, c_sm=d, c_sn=1, c_sz=d * d), bf16=bf16)
    ref = sum(_r(dPq[l], bf16).T @ _r(Qs[l], bf16) for l in range(L))
    assert _rel(part.sum(0), ref) < 2e-6


@pytest.mark.parametrize("a_m", [True, False])
def test_gemm_large_m(a_m):
    """Large M (several rounds of workgroups), partial edge tile."""
    torch.manual_seed(7)
    M, N, K = 50052, 256, 128
    A = torch.randn(K, M, device="cuda") if a_m else torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / 8
    bias = torch.randn(N, device="cuda")
    Cm = torch.full((M, N), float("nan"), device="cuda")
    kw = dict(A=A, B=W, C=Cm, bias_n=bias, M=M, N=N, K=K, batch=1, b_sk=1, b_sn=K, c_sm=N, c_sn=1)
    kw.update(dict(a_sm=1, a_sk=M) if a_m else dict(a_sm=K, a_sk=1))
    _gemm(kw)
    ref = (A.double().T if a_m else A.double()) @ W.double().T + bias.double()
    assert _rel(Cm, ref) < 2e-6


@pytest.mark.parametrize("a_m,b_n", [(True, True), (False, True), (True, False), (False, False)])
def test_gemm_small_tile_dispatch(a_m, b_n):
    """Launches with 769..2303 128-row tiles switch to 64-row tiles (all four staging layouts)."""
    torch.manual_seed(8)
    Z, M, N, K = 4, 2048, 2048, 64                  # 16 x 16 x 4 = 1024 tiles of 128 rows
    A = torch.randn(Z, K, M, device="cuda") if a_m else torch.randn(Z, M, K, device="cuda")
    Bm = torch.randn(Z, K, N, device="cuda") if b_n else torch.randn(Z, N, K, device="cuda")
    Cm = torch.full((Z, M, N), float("nan"), device="cuda")
    kw = dict(A=A, B=Bm, C=Cm, M=M, N=N, K=K, batch=Z, a_sz=M * K, b_sz=N * K, c_sz=M * N, c_sm=N, c_sn=1)
    kw.update(dict(a_sm=1, a_sk=M) if a_m else dict(a_sm=K, a_sk=1))
    kw.update(dict(b_sk=N, b_sn=1) if b_n else dict(b_sk=1, b_sn=K))
    _gemm(kw)
    ref = (A.double().transpose(1, 2) if a_m else A.double()) @ (Bm.double() if b_n else Bm.double().transpose(1, 2))
    assert _rel(Cm, ref) < 2e-6


# ---- coattn_linear_forward: nn.Linear against a pre-split weight (gemm_w.hip) ----------------------------------
def _linear(x, ld, W, bias, M, N, K, scale=0.0, flags=0, wimg=None):
    from vqa_amd import _lib
    lib = _lib.load()
    y = torch.full((M, N), float("nan"), device="cuda")
    if wimg is None:
        wimg = torch.empty(lib.coattn_linear_workspace_bytes(N, K) // 4, device="cuda")
    rc = lib.coattn_linear_forward(x.data_ptr(), ld, W.data_ptr(), bias.data_ptr() if bias is not None else None,
                                   y.data_ptr(), wimg.data_ptr(), M, N, K, scale, flags,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, y, wimg


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (4160, 512, 512), (1000, 96, 64), (129, 200, 160), (31360, 512, 512)])
def test_linear_presplit_weight(M, N, K):
    from vqa_amd import _lib
    torch.manual_seed(11)
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    rc, y, _ = _linear(x, K, W, b, M, N, K)
    _lib.check(rc, "coattn_linear_forward")
    ref = x.double() @ W.double().t() + b.double()
    assert _rel(y, ref) < 2e-6            # fp32 accuracy: the exact 3-way split, fp32 accumulation


def test_linear_strided_rows_scale_and_image_reuse():
    """Rows ld_x > K apart, out_scale, no bias; a second call reuses the weight image (flags bit 0)."""
    from vqa_amd import _lib
    torch.manual_seed(12)
    M, N, K, ld = 777, 256, 96, 128
    xs = torch.randn(M, ld, device="cuda")
    W = torch.randn(N, K, device="cuda")
    rc, y, wimg = _linear(xs, ld, W, None, M, N, K, scale=2.5)
    _lib.check(rc, "coattn_linear_forward")
    ref = 2.5 * (xs[:, :K].double() @ W.double().t())
    assert _rel(y, ref) < 2e-6
    x2 = torch.randn(M, ld, device="cuda")
    rc, y2, _ = _linear(x2, ld, torch.zeros_like(W), None, M, N, K, scale=2.5, flags=1, wimg=wimg)   # W ignored: image reused
    _lib.check(rc, "coattn_linear_forward")
    assert _rel(y2, 2.5 * (x2[:, :K].double() @ W.double().t())) < 2e-6


@pytest.mark.parametrize("M,N,K", [(31360, 512, 512), (4160, 512, 512), (777, 256, 96), (130, 96, 64)])
def test_linear_two_fp16_pieces(M, N, K):
    """COATTN_FLAG_F16PAIR (the form coattn_forward runs its projections in): two FP16 pieces per operand, the weight image
    scaled by 256 -- 22 significand bits; values of ordinary magnitude, and a weight image reused under the same flag."""
    from vqa_amd import _lib
    torch.manual_seed(13)
    x = torch.randn(M, K, device="cuda").clamp_min_(0) * 3.0          # post-ReLU-like features
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    rc, y, wimg = _linear(x, K, W, b, M, N, K, flags=_lib.FLAG_F16PAIR)
    _lib.check(rc, "coattn_linear_forward")
    ref = x.double() @ W.double().t() + b.double()
    assert _rel(y, ref) < 4e-6
    rc, y2, _ = _linear(x, K, torch.zeros_like(W), b, M, N, K, flags=_lib.FLAG_F16PAIR | 1, wimg=wimg)
    _lib.check(rc, "coattn_linear_forward")
    assert torch.equal(y, y2)


@pytest.mark.parametrize("name,M,N,K,flag", [
    ("gemm_bf: LDS-DMA weight ring, config 4's P_v shape", 7840, 2048, 2048, "BF16_PROJ"),
    ("gemm_h2: LDS-DMA weight ring of three buffers (K not 512: one workgroup per tile)", 31360, 512, 1024, "F16PAIR"),
    ("gemm_h2p: persistent pipeline, deferred stores, register weight ring (config 2's P_v shape)", 31360, 512, 512, "F16PAIR"),
    ("gemm_h2p at the step's own grid (7 x 7: two tiles per workgroup for some, one for others)", 7840, 512, 512, "F16PAIR")],
    ids=["gemm_bf_cfg4", "gemm_h2_dma", "gemm_h2p_n196", "gemm_h2p_n49"])
def test_ring_kernels_are_repeatable_and_right(name, M, N, K, flag):
    """The kernels whose operand rings are ordered by hand-counted waits (LDS-DMA refills behind read-backs, deferred stores
    in the vmcnt queue; DESIGN.md "LDS-DMA rings"): 25 launches on the same inputs into NaN-filled outputs, every one
    bit-identical to the first, and the first right against the float64 product (a race shows as a rare different tile --
    round 4 met one in the forward kernel's ring only at B = 640; VERDICT r4 asks for the sweep over gemm_bf as well)."""
    from vqa_amd import _lib
    torch.manual_seed(21)
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    fl = _lib.FLAG_BF16_PROJ if flag == "BF16_PROJ" else _lib.FLAG_F16PAIR
    rc, y0, wimg = _linear(x, K, W, b, M, N, K, flags=fl)
    _lib.check(rc, "coattn_linear_forward")
    rows = torch.arange(0, M, max(1, M // 997), device="cuda")
    if flag == "BF16_PROJ":
        ref = x[rows].bfloat16().double() @ W.bfloat16().double().t() + b.double()
        assert _rel(y0[rows], ref) < 1e-5, name
    else:
        ref = x[rows].double() @ W.double().t() + b.double()
        assert _rel(y0[rows], ref) < 4e-6, name
    for rep in range(25):
        rc, y, _ = _linear(x, K, W, b, M, N, K, flags=fl | (rep & 1), wimg=wimg)     # (every other call reuses the weight image)
        _lib.check(rc, "coattn_linear_forward")
        assert torch.equal(y, y0), (name, rep, int((y != y0).sum()))


def test_linear_two_fp16_pieces_range():
    """Beyond fp16's range the conversions saturate (MODE.FP16_OVFL): hi + lo carries magnitudes up to 131,008, larger ones
    clamp there -- finite results for any finite input; tiny values keep 2^-24 absolute."""
    from vqa_amd import _lib
    torch.manual_seed(14)
    M, N, K = 256, 128, 64
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    x = torch.randn(M, K, device="cuda")
    x[0, :] = 60000.0; x[1, :] = 1.0e5; x[2, :] = 1.0e9; x[3, :] = -3.0e38; x[4, :] = 1.0e-6; x[5, :] = 3.0e-8
    rc, y, _ = _linear(x, K, W, None, M, N, K, flags=_lib.FLAG_F16PAIR)
    _lib.check(rc, "coattn_linear_forward")
    assert torch.isfinite(y).all()
    ref = x.double() @ W.double().t()
    assert _rel(y[0:1], ref[0:1]) < 4e-6 and _rel(y[6:], ref[6:]) < 4e-6          # inside the range: 22 bits
    assert _rel(y[1:2], ref[1:2]) < 1e-3                                           # 65,504 .. 131,008: fewer bits
    clamped = torch.full((1, K), 131008.0, device="cuda").double() @ W.double().t()
    assert _rel(y[2:3], clamped) < 1e-3 and _rel(y[3:4], -clamped) < 1e-3          # beyond: clamped, finite
    assert (y[4:6].double() - ref[4:6]).abs().max() < 1e-6                          # tiny values: absolute 2^-24 per element


def test_linear_rejects_unsupported_shapes():
    from vqa_amd import _lib
    x = torch.randn(64, 48, device="cuda"); W = torch.randn(32, 48, device="cuda")
    rc, _, _ = _linear(x, 48, W, None, 64, 32, 48)          # K % 32 != 0, M < 128
    assert rc == -1 and b"not supported" in _lib.load().coattn_last_error()


# ---- coattn_linear_weight_grad: dW = dY^T X on the split-K A^T B kernel (gemm_tn.hip) --------------------------
def _wgrad(dy, ld_dy, x, ld_x, M, n_out, n_in, accumulate=0, dW=None):
    from vqa_amd import _lib
    lib = _lib.load()
    if dW is None:
        dW = torch.full((n_out, n_in), float("nan"), device="cuda")
    ws = torch.empty(lib.coattn_linear_wgrad_workspace_bytes(n_out, n_in) // 4, device="cuda")
    rc = lib.coattn_linear_weight_grad(dy.data_ptr(), ld_dy, x.data_ptr(), ld_x, dW.data_ptr(), ws.data_ptr(), M, n_out,
                                       n_in, accumulate, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, dW


@pytest.mark.parametrize("M,n_out,n_in", [(16, 128, 128), (17, 128, 256), (1000, 256, 128), (4160, 512, 512),
                                          (31360, 512, 512), (12345, 128, 384)])
def test_linear_weight_grad(M, n_out, n_in):
    """Contraction lengths that end inside a 16-row step and inside a split-K part; one to many parts."""
    from vqa_amd import _lib
    torch.manual_seed(21)
    dy = torch.randn(M, n_out, device="cuda") * 0.1
    x = torch.randn(M, n_in, device="cuda")
    rc, dW = _wgrad(dy, n_out, x, n_in, M, n_out, n_in)
    _lib.check(rc, "coattn_linear_weight_grad")
    ref = dy.double().t() @ x.double()
    assert _rel(dW, ref) < 2e-6


def test_linear_weight_grad_strided_accumulate_and_repeatable():
    from vqa_amd import _lib
    torch.manual_seed(22)
    M, n_out, n_in, ld_dy, ld_x = 3001, 128, 256, 160, 320
    dy = torch.randn(M, ld_dy, device="cuda"); x = torch.randn(M, ld_x, device="cuda")
    base = torch.randn(n_out, n_in, device="cuda")
    rc, dW = _wgrad(dy, ld_dy, x, ld_x, M, n_out, n_in, accumulate=1, dW=base.clone())
    _lib.check(rc, "coattn_linear_weight_grad")
    ref = base.double() + dy[:, :n_out].double().t() @ x[:, :n_in].double()
    assert _rel(dW, ref) < 2e-6
    rc, dW2 = _wgrad(dy, ld_dy, x, ld_x, M, n_out, n_in, accumulate=1, dW=base.clone())
    assert torch.equal(dW, dW2)                              # fixed split-K partition and reduction order


def test_linear_weight_grad_rejects_unsupported_shapes():
    from vqa_amd import _lib
    dy = torch.randn(64, 96, device="cuda"); x = torch.randn(64, 128, device="cuda")
    rc, _ = _wgrad(dy, 96, x, 128, 64, 96, 128)               # n_out not a multiple of 128
    assert rc == -1 and b"not supported" in _lib.load().coattn_last_error()


# ---- the reduced-precision mode (COATTN_FLAG_BF16_PROJ) of the linear entry points -----------------------------------
# operands rounded to bf16 (round to nearest even), one MFMA per product, fp32 accumulation: against the float64
# product of the ROUNDED operands only the accumulation order differs (tolerance 2e-5 of max|.|)
@pytest.mark.parametrize("M,N,K", [(7840, 2048, 2048), (300, 256, 64), (1000, 512, 192), (4160, 512, 512), (129, 200, 160)],
                         ids=lambda v: str(v))
def test_linear_reduced_precision(M, N, K):
    """Wide shapes run on gemm_bf.hip (256 x 256 tiles, LDS-DMA weight image; ragged last row tile, one to many k
    steps), the others on gemm_w.hip's single-piece mode."""
    from vqa_amd import _lib
    torch.manual_seed(31)
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    rc, y, _ = _linear(x, K, W, b, M, N, K, flags=_lib.FLAG_BF16_PROJ)
    _lib.check(rc, "coattn_linear_forward")
    ref = x.bfloat16().double() @ W.bfloat16().double().t() + b.double()
    assert _rel(y, ref) < 2e-5
    assert _rel(y, x.double() @ W.double().t() + b.double()) > 1e-4        # it IS the reduced-precision product


@pytest.mark.parametrize("M,n_out,n_in", [(7840, 2048, 2048), (64, 256, 256), (4160, 512, 768), (12345 // 32 * 32, 256, 512),
                                          (1000, 256, 128)], ids=lambda v: str(v))
def test_linear_weight_grad_reduced_precision(M, n_out, n_in):
    """dW = dY^T X in the same mode: gemm_bf.hip's 256 x 256 split-K kernel where the shape allows (rows % 32 == 0,
    256-multiples), gemm_tn.hip's single-piece mode otherwise; repeatable bit for bit."""
    from vqa_amd import _lib
    torch.manual_seed(32)
    dy = torch.randn(M, n_out, device="cuda") * 0.1
    x = torch.randn(M, n_in, device="cuda")
    rc, dW = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_PROJ)
    _lib.check(rc, "coattn_linear_weight_grad")
    ref = dy.bfloat16().double().t() @ x.bfloat16().double()
    assert _rel(dW, ref) < 2e-5
    rc, dW2 = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_PROJ)
    assert torch.equal(dW, dW2)
    base = torch.ones(n_out, n_in, device="cuda")
    rc, dW3 = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_PROJ | 1, dW=base.clone())
    assert _rel(dW3 - 1.0, ref) < 5e-5


@pytest.mark.parametrize("M,N,K", [(12480, 2048, 2048), (300, 256, 64), (1000, 512, 192)], ids=lambda v: str(v))
def test_linear_bf16_stored_activations(M, N, K):
    """COATTN_FLAG_BF16_IN: x is STORED as bf16 (an autocast encoder's activations, the fused backward's own dP_q) and
    read as it is -- bit for bit the result of the fp32-stored path fed the same (rounded) values; unsupported shapes
    are refused."""
    from vqa_amd import _lib
    torch.manual_seed(33)
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    xb = x.bfloat16().contiguous()
    rc, y, _ = _linear(xb, K, W, b, M, N, K, flags=_lib.FLAG_BF16_PROJ | _lib.FLAG_BF16_IN)
    _lib.check(rc, "coattn_linear_forward")
    rc, y32, _ = _linear(xb.float(), K, W, b, M, N, K, flags=_lib.FLAG_BF16_PROJ)
    _lib.check(rc, "coattn_linear_forward")
    assert torch.equal(y, y32)
    assert _rel(y, xb.double() @ W.bfloat16().double().t() + b.double()) < 2e-5
    rc, _, _ = _linear(xb, K, W, b, M, N, K, flags=_lib.FLAG_BF16_IN)            # without the reduced-precision mode
    assert rc != 0
    rc, _, _ = _linear(xb[:128], K, W, b, 128, N, K, flags=_lib.FLAG_BF16_PROJ | _lib.FLAG_BF16_IN)   # not a gemm_bf shape
    assert rc != 0


@pytest.mark.parametrize("M,n_out,n_in", [(7840, 2048, 2048), (64, 256, 256), (12345 // 32 * 32, 256, 512)], ids=lambda v: str(v))
def test_linear_weight_grad_bf16_stored_gradients(M, n_out, n_in):
    """The same for dW = dY^T X with dY stored as bf16 (what bwd_nat32 writes in the reduced-precision mode)."""
    from vqa_amd import _lib
    torch.manual_seed(34)
    dy = (torch.randn(M, n_out, device="cuda") * 0.1).bfloat16().contiguous()
    x = torch.randn(M, n_in, device="cuda")
    rc, dW = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_PROJ | _lib.FLAG_BF16_IN)
    _lib.check(rc, "coattn_linear_weight_grad")
    rc, dW32 = _wgrad(dy.float(), n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_PROJ)
    _lib.check(rc, "coattn_linear_weight_grad")
    assert torch.equal(dW, dW32)
    assert _rel(dW, dy.double().t() @ x.bfloat16().double()) < 2e-5
    rc, _ = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_BF16_IN)
    assert rc != 0


# ---- the two-piece width (COATTN_FLAG_SPLIT2) of the linear entry points -------------------------------------------
# operands truncated to their first two bf16 pieces (hi + mid, round to nearest at each step), the three partial
# products hi*mid, mid*hi, hi*hi, fp32 accumulation: against the float64 value of exactly those products only the
# accumulation order differs (2e-6 of max|.|); against the true product the error is ~2^-16 per product, random in sign
def _two_piece(x):
    h = x.bfloat16().float()
    m = (x - h).bfloat16().float()
    return h.double(), m.double()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (4160, 512, 512), (1000, 96, 64), (129, 200, 160), (31360, 512, 512)])
def test_linear_two_piece_width(M, N, K):
    from vqa_amd import _lib
    torch.manual_seed(41)
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    rc, y, _ = _linear(x, K, W, b, M, N, K, flags=_lib.FLAG_SPLIT2)
    _lib.check(rc, "coattn_linear_forward")
    xh, xm = _two_piece(x)
    wh, wm = _two_piece(W)
    ref2 = (xh + xm) @ wh.t() + xh @ wm.t() + b.double()
    assert _rel(y, ref2) < 2e-6
    true = x.double() @ W.double().t() + b.double()
    e = _rel(y, true)
    print("two-piece linear %s: %.2e of max|.| from the true product" % ((M, N, K), e))
    assert 1e-7 < e < 3e-5                                  # it IS the two-piece product, and no worse than its budget


@pytest.mark.parametrize("M,n_out,n_in", [(17, 128, 256), (4160, 512, 512), (31360, 512, 512), (12345, 128, 384)])
def test_linear_weight_grad_two_piece_width(M, n_out, n_in):
    from vqa_amd import _lib
    torch.manual_seed(42)
    dy = torch.randn(M, n_out, device="cuda") * 0.1
    x = torch.randn(M, n_in, device="cuda")
    rc, dW = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_SPLIT2)
    _lib.check(rc, "coattn_linear_weight_grad")
    dh, dm = _two_piece(dy)
    xh, xm = _two_piece(x)
    ref2 = (dh + dm).t() @ xh + dh.t() @ xm
    assert _rel(dW, ref2) < 2e-6
    e = _rel(dW, dy.double().t() @ x.double())
    print("two-piece weight grad %s: %.2e of max|.| from the true product" % ((M, n_out, n_in), e))
    assert e < 3e-5
    rc, dW2 = _wgrad(dy, n_out, x, n_in, M, n_out, n_in, accumulate=_lib.FLAG_SPLIT2)
    assert torch.equal(dW, dW2)
