import torch, time
dev="cuda"
R,H=150*4096*8,256
x=torch.randn(R,H,device=dev); W=torch.randn(H,H,device=dev)*0.05; Wt=W.t().contiguous()
def t(fn,it=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/it*1e3
fl=2*R*H*H/1e12
import torch.nn.functional as F
out=torch.empty(R,H,device=dev)
cases={
 "F.linear(x,W)            ": lambda: F.linear(x,W),
 "x @ Wt (contig)          ": lambda: torch.mm(x,Wt),
 "x @ W.t() (view)         ": lambda: torch.mm(x,W.t()),
 "mm out=                  ": lambda: torch.mm(x,Wt,out=out),
 "bmm 128 chunks, W.t()    ": lambda: torch.bmm(x.view(128,R//128,H), W.t().expand(128,H,H)),
 "bmm 128 chunks, Wt contig": lambda: torch.bmm(x.view(128,R//128,H), Wt.expand(128,H,H)),
 "bmm 16 chunks            ": lambda: torch.bmm(x.view(16,R//16,H), Wt.expand(16,H,H)),
 "(W @ x.t()).t()          ": lambda: torch.mm(W,x.t()),
 "dgrad-like dz @ W        ": lambda: torch.mm(x,W),
}
for k,f in cases.items():
    ms=t(f); print("%s %.2f ms  %.1f TF/s"%(k,ms,fl/ms*1e3))
torch.backends.cuda.preferred_blas_library("hipblaslt")
print("preferred hipblaslt"); 
for k in list(cases)[:2]:
    ms=t(cases[k]); print("%s %.2f ms  %.1f TF/s"%(k,ms,fl/ms*1e3))
try:
    torch.backends.cuda.preferred_blas_library("cublas")
    print("preferred rocblas")
    for k in list(cases)[:3]:
        ms=t(cases[k]); print("%s %.2f ms  %.1f TF/s"%(k,ms,fl/ms*1e3))
except Exception as e: print(e)
