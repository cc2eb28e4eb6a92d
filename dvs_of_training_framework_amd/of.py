"""``OpticalFlow`` inference wrapper with the reference's contract
(DummyNet/of.py:17-125, used by test.py:51-60): events of several windows in,
``np.ndarray [B,H,W,2]`` out."""
from os import path as osp

import numpy as np
import torch
import torch.nn as nn

from .net import Model

script_dir = osp.dirname(osp.realpath(__file__))


class OpticalFlow:
    def __init__(self, imsize, model=osp.join(script_dir, 'data/model/model.pth'),
                 device=torch.device('cuda:0'), activation=nn.ReLU(),
                 **model_kwargs):
        self._device = torch.device(device)
        self._net = Model(device=self._device, activation=activation,
                          **model_kwargs)
        if model is not None:
            state_dict = torch.load(model, map_location=self._device,
                                    weights_only=True)
            if 'model' in state_dict:
                state_dict = state_dict['model']
            self.load_state_dict(state_dict)
        self._net.eval()
        self.imsize = imsize

    def load_state_dict(self, state_dict):
        self._net.load_state_dict(state_dict)
        self._net.to(device=self._device)

    def __call__(self, events, start, stop, return_all=False):
        """events: per sample (x, y, t, p) iterables, p in {-1, 1};
        start/stop: window bounds per sample."""
        with torch.no_grad():
            flow, _, _ = self._net(*self._collate(events, start, stop),
                                   self.imsize)
            return self._postprocess(flow, return_all)

    def _collate(self, events, start, stop):
        """Stack the samples into one event dict with a sample index
        (DummyNet/of.py:76-115); timestamps are shifted to start at 0."""
        cols = {k: [] for k in ('x', 'y', 'timestamp', 'polarity',
                                'element_index', 'sample_index')}
        for i, e in enumerate(events):
            x, y, t, p = (np.asarray(v) for v in e[:4])
            cols['x'].append(x.astype(np.int64))
            cols['y'].append(y.astype(np.int64))
            cols['timestamp'].append(t.astype(np.float64))
            cols['polarity'].append(p.astype(np.int64))
            cols['element_index'].append(np.zeros(x.size, np.int64))
            cols['sample_index'].append(np.full(x.size, i, np.int64))
        timestamps = np.hstack([[b, e] for b, e in zip(start, stop)]) \
            .astype(np.float64)
        sample_idx = np.hstack([[i, i] for i in range(len(start))])
        min_t = timestamps.min()
        ev = {}
        for k, v in cols.items():
            a = np.concatenate(v) if v else np.zeros(0)
            if k == 'timestamp':
                ev[k] = torch.tensor(a - min_t, dtype=torch.float32,
                                     device=self._device)
            else:
                ev[k] = torch.tensor(a, dtype=torch.long, device=self._device)
        return ev, \
            torch.tensor(timestamps - min_t, dtype=torch.float32,
                         device=self._device), \
            torch.tensor(sample_idx, dtype=torch.long, device=self._device)

    def _postprocess(self, flow, return_all):
        def back(f):
            return np.transpose(f.detach().cpu().numpy(), (0, 2, 3, 1))
        if return_all:
            return tuple(map(back, flow))
        return back(flow[-1])
