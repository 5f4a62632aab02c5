import sys, os, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd
from vqa_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
def run(name, kw, flop, iters=30):
    g = _lib.GemmDesc()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor): keep.append(v); v = v.data_ptr()
        setattr(g, k, v)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): _lib.check(lib.coattn_gemm_f32(C.byref(g), st), name)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): lib.coattn_gemm_f32(C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print("%-10s %8.1f us  %6.1f TFLOP/s" % (name, t * 1e6, flop / t / 1e12))
B, N, T, d = 160, 196, 26, 512
V = torch.randn(B, d, N, device=dev); W = torch.randn(d, d, device=dev); Pv = torch.empty(B * N, d, device=dev)
run("P_v", dict(A=V, B=W, C=Pv, M=B*N, N=d, K=d, batch=1, a_sm=1, a_sk=N, a_mdiv=N, a_sdiv=d*N, b_sk=1, b_sn=d, c_sm=d, c_sn=1), 2.0*B*N*d*d)
Q = torch.randn(3 * B * T, d, device=dev); Pq = torch.empty(3 * B * T, d, device=dev)
run("P_q(x3)", dict(A=Q, B=W, C=Pq, M=3*B*T, N=d, K=d, batch=1, a_sm=d, a_sk=1, b_sk=1, b_sn=d, c_sm=d, c_sn=1), 2.0*3*B*T*d*d)
run("dQproj", dict(A=Q, B=W, C=Pq, M=3*B*T, N=d, K=d, batch=1, a_sm=d, a_sk=1, b_sk=d, b_sn=1, c_sm=d, c_sn=1), 2.0*3*B*T*d*d)
G = 5; S = 32
part = torch.empty(S, d, d, device=dev)
run("dW_v", dict(A=Pv, B=V, C=part, M=d, N=d, K=N, batch=S, inner=G, inner_total=B, a_sm=1, a_sk=d, a_si=N*d, a_sz=G*N*d,
                 b_sk=1, b_sn=N, b_si=d*N, b_sz=G*d*N, c_sm=d, c_sn=1, c_sz=d*d), 2.0*B*N*d*d)
K = 3 * B * T; ks = (K // 32 + 15) // 16 * 16; S2 = (K + ks - 1) // ks
part2 = torch.empty(S2, d, d, device=dev)
run("dW_q", dict(A=Pq, B=Q, C=part2, M=d, N=d, K=K, batch=S2, ksplit=ks, a_sm=1, a_sk=d, b_sk=d, b_sn=1, c_sm=d, c_sn=1, c_sz=d*d), 2.0*K*d*d)
# the same products on location-major features (channels_last encoder): plain row-major P_v, flat split-K dW_v
Vl = torch.randn(B * N, d, device=dev)
run("P_v lm", dict(A=Vl, B=W, C=Pv, M=B*N, N=d, K=d, batch=1, a_sm=d, a_sk=1, b_sk=1, b_sn=d, c_sm=d, c_sn=1), 2.0*B*N*d*d)
Kv = B * N; ksv = ((Kv + 31) // 32 + 15) // 16 * 16; Sv = (Kv + ksv - 1) // ksv
partv = torch.empty(Sv, d, d, device=dev)
run("dW_v lm", dict(A=Pv, B=Vl, C=partv, M=d, N=d, K=Kv, batch=Sv, ksplit=ksv, a_sm=1, a_sk=d, b_sk=d, b_sn=1, c_sm=d, c_sn=1, c_sz=d*d), 2.0*Kv*d*d)
if os.environ.get("SKIP_TORCH"): sys.exit(0)
# practical ceiling: the vendor library's fp32 GEMM on the same shapes (contiguous operands)
def tref(name, fn, flop, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print("%-22s %8.1f us  %6.1f TFLOP/s" % (name, t * 1e6, flop / t / 1e12))
X = torch.randn(B * N, d, device=dev); out = torch.empty(B * N, d, device=dev)
tref("torch P_v-shape NT", lambda: torch.mm(X, W.t(), out=out), 2.0*B*N*d*d)
tref("torch P_v-shape NN", lambda: torch.mm(X, W, out=out), 2.0*B*N*d*d)
Xt = torch.randn(d, B * N, device=dev); o2 = torch.empty(d, d, device=dev)
tref("torch dW_v-shape TN", lambda: torch.mm(Xt, X, out=o2), 2.0*B*N*d*d)
A4 = torch.randn(4096, 4096, device=dev); B4 = torch.randn(4096, 4096, device=dev); o4 = torch.empty(4096, 4096, device=dev)
tref("torch 4096^3", lambda: torch.mm(A4, B4, out=o4), 2.0*4096**3)
Vb = torch.randn(B, d, N, device=dev); ob = torch.empty(B, N, d, device=dev)
tref("torch bmm V^T.W^T (in place layout)", lambda: torch.matmul(Vb.transpose(1, 2), W.t(), out=ob), 2.0*B*N*d*d)
