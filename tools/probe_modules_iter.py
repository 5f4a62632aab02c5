import sys, time; sys.path.insert(0,'/root/repo')
import torch, vqa_amd
from vqa_amd.modules import MLPClassifier
import bench
dev=torch.device('cuda',0)
B,T,d,K=160,26,512,1000
for N in (196, 49):
    torch.manual_seed(0)
    co=vqa_amd.ParallelCoAttention(d).to(dev); mlp=MLPClassifier(d,1024,K+1).to(dev)
    V,Qs=bench.synth_features(B,N,T,d,dev)
    x=V.permute(0,2,1).contiguous(); Qs=[q.requires_grad_(True) for q in Qs]
    label=(torch.arange(B,device=dev)*7)%(K+1)
    params=list(co.parameters())+list(mlp.parameters())
    def mstep():
        for p in params: p.grad=None
        _,loss=mlp.forward_loss(*co(x,Qs),label); loss.backward()
    ts=[]
    for i in range(40):
        torch.cuda.synchronize(); t0=time.perf_counter(); mstep(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    print(N, ' '.join('%.2f'%t for t in ts))
