"""Developer probe for counter passes: 30 calls of coattn_linear_forward (two FP16 pieces) at M = 31360, N = K = 512."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "31360")); d = 512
x = torch.randn(M, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
y = torch.empty(M, d, device=dev); wimg = torch.empty(lib.coattn_linear_workspace_bytes(d, d) // 4, device=dev)
lib.coattn_linear_forward(x.data_ptr(), d, W.data_ptr(), None, y.data_ptr(), wimg.data_ptr(), M, d, d, 0.0, _lib.FLAG_F16PAIR, st)
for _ in range(30):
    lib.coattn_linear_forward(x.data_ptr(), d, W.data_ptr(), None, y.data_ptr(), wimg.data_ptr(), M, d, d, 0.0, _lib.FLAG_F16PAIR | 1, st)
torch.cuda.synchronize()
