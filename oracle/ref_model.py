"""TEST INFRASTRUCTURE ONLY (see oracle/dvsof_oracle.c header).

Plain PyTorch fp32 restatement of docs/MODEL_SPEC.md (CPU or GPU, ATen ops
only) -- the floating-point reference the HIP conv stack is compared with.
Upstream's EV_FlowNet source is absent ("parity unpinned"), so this
file and the spec define the network; it takes the SAME state_dict as
dvs_of_training_framework_amd.predictor.Predictor."""
import torch
import torch.nn.functional as F


def _act(x, mish):
    return F.mish(x) if mish else F.relu(x)


def ref_predictor(state, x, mish=False, prefix=''):
    """state: dict name -> tensor (names as Predictor.state_dict()).
    x: [B,C,H,W].  -> list of 4 flows coarse to fine."""
    g = lambda n: state[prefix + n]
    e = []
    for i in range(4):
        x = _act(F.conv2d(x, g(f'enc.{i}.conv.weight'), g(f'enc.{i}.conv.bias'),
                          stride=2, padding=1), mish)
        e.append(x)
    r = x
    for i in range(2):
        t = _act(F.conv2d(r, g(f'res.{i}.conv1.weight'),
                          g(f'res.{i}.conv1.bias'), padding=1), mish)
        r = _act(F.conv2d(t, g(f'res.{i}.conv2.weight'),
                          g(f'res.{i}.conv2.bias'), padding=1) + r, mish)
    flows, x, f = [], r, None
    for i in range(4):
        parts = [x, e[3 - i]] + ([f] if f is not None else [])
        inp = F.interpolate(torch.cat(parts, 1), scale_factor=2, mode='nearest')
        x = _act(F.conv2d(inp, g(f'dec.{i}.conv.weight'),
                          g(f'dec.{i}.conv.bias'), padding=1), mish)
        f = F.conv2d(x, g(f'dec.{i}.flow.weight'), g(f'dec.{i}.flow.bias'))
        flows.append(f)
    return flows
