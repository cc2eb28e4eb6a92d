"""TEST INFRASTRUCTURE ONLY (see oracle/dvsof_oracle.c header).

Plain PyTorch (CPU, fp32) restatements of the optimizers whose sources are
un-vendored submodules upstream ("parity unpinned"): RAdam (Liu et al., "On the
Variance of the Adaptive Learning Rate and Beyond", ICLR 2020, reference
implementation's update rule) and Ranger (lessw2020: RAdam + Lookahead
[Zhang et al. 2019] + gradient centralisation [Yong et al. 2020])."""
import math

import torch


def _rect(step, beta1, beta2):
    b2t = beta2 ** step
    nmax = 2.0 / (1.0 - beta2) - 1.0
    nsma = nmax - 2.0 * step * b2t / (1.0 - b2t)
    bc1 = 1.0 - beta1 ** step
    adaptive = math.sqrt((1 - b2t) * (nsma - 4) / (nmax - 4) * (nsma - 2) / nsma *
                         nmax / (nmax - 2)) / bc1 if nsma > 4 else None
    return nsma, adaptive, 1.0 / bc1


class RefRAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, degenerated_to_sgd=True):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.degenerated = degenerated_to_sgd
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        nsma, adaptive, sgd = _rect(self.t, b1, b2)
        for p, m, v in zip(self.params, self.m, self.v):
            g = p.grad
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            m.mul_(b1).add_(g, alpha=1 - b1)
            if nsma >= 5:
                if self.wd:
                    p.add_(p, alpha=-self.wd * self.lr)
                p.addcdiv_(m, v.sqrt().add_(self.eps), value=-adaptive * self.lr)
            elif self.degenerated:
                if self.wd:
                    p.add_(p, alpha=-self.wd * self.lr)
                p.add_(m, alpha=-sgd * self.lr)


class RefRanger:
    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5,
                 betas=(.95, 0.999), eps=1e-5, weight_decay=0.0, use_gc=True,
                 gc_conv_only=False):
        self.params = list(params)
        self.lr, self.alpha, self.k, self.thr = lr, alpha, k, N_sma_threshhold
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.gc_dim = (3 if gc_conv_only else 1) if use_gc else None
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.slow = [p.detach().clone() for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        nsma, adaptive, sgd = _rect(self.t, b1, b2)
        for p, m, v, slow in zip(self.params, self.m, self.v, self.slow):
            g = p.grad.clone()
            if self.gc_dim is not None and g.dim() > self.gc_dim:
                g.add_(-g.mean(dim=tuple(range(1, g.dim())), keepdim=True))
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            m.mul_(b1).add_(g, alpha=1 - b1)
            if self.wd:
                p.add_(p, alpha=-self.wd * self.lr)
            if nsma > self.thr:
                p.addcdiv_(m, v.sqrt().add_(self.eps), value=-adaptive * self.lr)
            else:
                p.add_(m, alpha=-sgd * self.lr)
            if self.t % self.k == 0:
                slow.add_(p - slow, alpha=self.alpha)
                p.copy_(slow)
