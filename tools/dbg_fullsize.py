import sys, torch
sys.path.insert(0,'/root/repo')
from dvs_of_training_framework_amd.predictor import Predictor
from oracle.ref_model import ref_predictor
torch.manual_seed(2)
B,Cin,H,W=8,5,256,256
net=Predictor(Cin).cuda(); x=torch.randn(B,Cin,H,W,device='cuda')
gfl=[torch.randn(B,2,H//s,W//s,device='cuda')*(0.5/s) for s in (8,4,2,1)]
flows=net(x); torch.autograd.backward(flows,gfl)
state={k:v.detach().clone().contiguous().requires_grad_(True) for k,v in net.state_dict().items()}
ref=ref_predictor(state,x); torch.autograd.backward(ref,gfl)
# also CPU reference in double for arbitration
state64={k:v.detach().cpu().double().contiguous().requires_grad_(True) for k,v in net.state_dict().items()}
ref64=ref_predictor(state64,x.cpu().double()); torch.autograd.backward(ref64,[g.cpu().double() for g in gfl])
for n,p in net.named_parameters():
    g=p.grad; r=state[n].grad; r64=state64[n].grad.cuda().float()
    print(f'{n:22s} hip-vs-aten L2 {float((g-r).norm()/r.norm()):.2e} max {float((g-r).abs().max()/r.abs().max()):.2e} | hip-vs-f64 L2 {float((g-r64).norm()/r64.norm()):.2e} | aten-vs-f64 L2 {float((r-r64).norm()/r64.norm()):.2e}')
