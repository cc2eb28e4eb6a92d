import sys, torch
sys.path.insert(0, '/root/repo')
from dvs_of_training_framework_amd.predictor import Predictor
for mode, seed in (('bf16', 11), ('bf16x3', 12)):
    torch.manual_seed(seed)
    a = Predictor(5).cuda(); b = Predictor(5, compute_dtype=mode).cuda(); b.load_state_dict(a.state_dict())
    x = torch.randn(2, 5, 64, 64, device='cuda')
    fa, fb = a(x), b(x)
    seeds = [torch.randn_like(f) for f in fa]
    torch.autograd.backward(fa, seeds); torch.autograd.backward(fb, seeds)
    print(mode, 'flow rel', max(float((u - v).norm() / u.norm()) for u, v in zip(fa, fb)))
    rel = [float((p.grad - q.grad).norm() / (p.grad.norm() + 1e-12)) for p, q in zip(a.parameters(), b.parameters())]
    cos = [float((p.grad * q.grad).sum() / (p.grad.norm() * q.grad.norm() + 1e-12)) for p, q in zip(a.parameters(), b.parameters())]
    print(mode, 'grad rel max', max(rel), 'cos min', min(cos))
