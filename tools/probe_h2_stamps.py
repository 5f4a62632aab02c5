"""Developer probe (library built with -DGEMMH2_STAMPS=1, tools/ab_gemmh2.sh build GEMMH2_STAMPS=1): per-half-step shader-clock
stamps of wave 0 of every workgroup of the persistent projection kernel, read back from the weight image's unused third KB."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(os.environ.get("M", "31360")); d = 512
x = torch.randn(M, d, device=dev); W = torch.randn(d, d, device=dev) / d ** 0.5
y = torch.empty(M, d, device=dev); wimg = torch.zeros(lib.coattn_linear_workspace_bytes(d, d) // 4, device=dev)
call = lambda f: lib.coattn_linear_forward(x.data_ptr(), d, W.data_ptr(), None, y.data_ptr(), wimg.data_ptr(), M, d, d, 0.0, _lib.FLAG_F16PAIR | f, st)
call(0)
for _ in range(300): call(1)
torch.cuda.synchronize()
img = wimg.view(torch.int64).view(-1, 3 * 128)[:, 256:].cpu()      # [chunk][128 stamps]
nwg = min(256, img.shape[0])
import statistics
rows = []
clks = []
for b in range(nwg):
    t = img[b].tolist()
    n = max(i for i in range(126) if t[i] != 0) + 1 if any(t[:126]) else 0
    if n < 3: continue
    rows.append([t[i + 1] - t[i] for i in range(n - 1)] + [None] * (126 - n))
    clks.append((t[n - 1] - t[0]) / max(1, (t[126] - t[127])) * 100.0 if t[126] > t[127] else 0)
print("shader clock over the stamped span: median %.0f MHz (min %.0f, max %.0f)" % (statistics.median(clks), min(clks), max(clks)))
print("workgroups with stamps:", len(rows), "stamps per wg:", sum(1 for v in rows[0] if v is not None) + 1)
nh = sum(1 for v in rows[0] if v is not None)
med = [statistics.median(r[i] for r in rows if r[i] is not None) for i in range(nh)]
print("median cycles per half step (first = prologue -> first stamp):")
print(" ".join("%d" % m for m in med))
if os.environ.get("SLOTS"):
    print("per-slot (steps 4, 5 of each tile; 24 slots per step):")
    for k in range(0, nh, 24): print("   ", " ".join("%d" % m for m in med[k:k + 24]))
    sys.exit(0)
print("tile 1 half0 avg %.0f half1 avg %.0f | total median per wg %.0f cycles" % (
    statistics.mean(med[1:32:2]), statistics.mean(med[2:33:2]), statistics.median(sum(v for v in r if v is not None) for r in rows)))
