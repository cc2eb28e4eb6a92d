import sys, torch
sys.path.insert(0,'.')
from dvs_of_training_framework_amd import conv as C
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (B,H,W,Cc) in [(8,256,256,32),(8,128,128,64),(8,64,64,128),(8,32,32,256)]:
    x=torch.randn(B,H,W,Cc,device='cuda'); w=torch.randn(2,Cc,device='cuda'); b=torch.randn(2,device='cuda')
    out=torch.empty(B,2,H,W,device='cuda')
    import ctypes
    def f():
        C._lib.check(C._lib.lib().dvsof_flow_head_fwd(x.data_ptr(),w.data_ptr(),b.data_ptr(),out.data_ptr(),B,H,W,Cc,C._lib.stream()),'h')
    print(B,H,W,Cc,'%.1f us'%timeit(f))
print('head_bwd')
for (B,H,W,Cc) in [(8,256,256,32),(8,128,128,64),(8,64,64,128),(8,32,32,256)]:
    x=torch.randn(B,H,W,Cc,device='cuda'); w=torch.randn(2,Cc,device='cuda')
    gf=torch.randn(B,2,H,W,device='cuda'); gxin=torch.randn(B,H,W,Cc,device='cuda')
    gx=torch.empty(B,H,W,Cc,device='cuda'); dw=torch.empty(2,Cc,device='cuda'); db=torch.empty(2,device='cuda')
    def f():
        C.head_bwd(x,w,gf,gxin,x,C.ACT_RELU,gx,dw,db,B,H,W,Cc)
    print(B,H,W,Cc,'%.1f us'%timeit(f))
