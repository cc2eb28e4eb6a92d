"""A minimal plugin honouring the reference's Model contract
(DummyNet/net.py:42-80) used to test the loader and the train loop on CPU."""
import torch
from torch import nn


class Model(nn.Module):
    def __init__(self, device, prefix_length=0, suffix_length=0):
        super().__init__()
        self.prefix_length, self.suffix_length = prefix_length, suffix_length
        self.scale = nn.Parameter(torch.ones(1))
        self.to(device)

    def forward(self, events, timestamps, sample_idx, imsize, raw=True,
                intermediate=False):
        B = int(sample_idx[-1]) + 1
        T = timestamps.numel() // B
        flows = tuple(self.scale * torch.ones(B, 2, imsize[0] // 2 ** i,
                                               imsize[1] // 2 ** i)
                      for i in (3, 2, 1, 0))
        ts = timestamps.view(B, T)[:, self.prefix_length:self.prefix_length + 2]
        out = (flows, ts, sample_idx.view(B, T)[:, 0])
        return out + ((tuple(),) if intermediate else tuple())
