"""Developer probe: fixed cost of the strided GEMM kernels on small problems (M = 160 rows): time vs K."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vqa_amd  # noqa: E402
from vqa_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, N, K, a_kcontig, b_kcontig, batch=1, ksplit=0, iters=200):
    A = torch.randn(M, K, device=dev) if a_kcontig else torch.randn(K, M, device=dev)
    Bm = torch.randn(N, K, device=dev) if b_kcontig else torch.randn(K, N, device=dev)
    Cm = torch.empty(max(batch, 1), M, N, device=dev)
    g = _lib.GemmDesc()
    g.A, g.B, g.C = A.data_ptr(), Bm.data_ptr(), Cm.data_ptr()
    g.M, g.N, g.K, g.batch, g.ksplit = M, N, K, batch, ksplit
    g.a_sm, g.a_sk = (K, 1) if a_kcontig else (1, M)
    g.b_sk, g.b_sn = (1, K) if b_kcontig else (N, 1)
    g.c_sm, g.c_sn, g.c_sz = N, 1, M * N
    for _ in range(10):
        _lib.check(lib.coattn_gemm_f32(C.byref(g), stream), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.coattn_gemm_f32(C.byref(g), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, ak, bk in (("A k-contig, B k-contig (forward layer)", True, True), ("A k-contig, B n-contig (dX)", True, False),
                     ("A m-contig, B n-contig (dW)", False, False)):
    print(name)
    for (M, N, K, batch, ks) in ((160, 512, 64, 1, 0), (160, 512, 128, 1, 0), (160, 512, 256, 1, 0), (160, 512, 512, 1, 0),
                                 (160, 512, 512, 8, 64), (160, 1024, 1024, 16, 64), (1024, 1024, 160, 1, 0),
                                 (1024, 512, 160, 1, 0)):
        print("   M=%4d N=%4d K=%4d batch=%2d ksplit=%3d : %7.1f us" % (M, N, K, batch, ks, run(M, N, K, ak, bk, batch, ks)), flush=True)
