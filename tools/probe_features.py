import sys; sys.path.insert(0,'/root/repo')
import torch, vqa_amd
from vqa_amd.coattention import native_features
dev='cuda'
for (B,N,d) in ((160,196,512),(160,49,512),(160,49,2048)):
    x=torch.randn(B,d,N,device=dev).permute(0,2,1)
    out=torch.empty(B,N,d,device=dev)
    lib=vqa_amd._lib.load()
    import ctypes as C
    from vqa_amd import _lib
    def call():
        sB,sN,sD=x.stride()
        lib.coattn_features_native(C.c_void_p(x.data_ptr()), _lib.F32, sB,sN,sD, C.c_void_p(out.data_ptr()), B,N,d, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(50): call()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/200*1e3
    e0.record()
    for _ in range(200): out.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t2=e0.elapsed_time(e1)/200*1e3
    print("B%d N%d d%d: features_native %.1f us (%.2f TB/s), torch copy_ %.1f us" % (B,N,d,t, 2*B*N*d*4/t/1e6, t2))
