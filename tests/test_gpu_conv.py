"""HIP conv stack (through the C ABI) against plain PyTorch fp32 CPU ops.
Tolerance: 1e-3 relative to the tensor's peak magnitude (north star: flow
fields within 1e-3 relative fp32); typical error is ~1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def close(got, want, rtol=RTOL):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= rtol * ref + 1e-7, (err, ref)


def nhwc(t):   # logical NCHW -> dense NHWC buffer on the GPU
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def from_nhwc(t):
    return t.permute(0, 3, 1, 2)


def wphys(w):  # OIHW -> [O][kh][kw][I] on the GPU
    return w.permute(0, 2, 3, 1).contiguous().cuda()


CASES = [
    # B, H, W, chans(list, layouts), Cout, k, stride, up, act
    dict(B=2, H=16, W=16, src=[(5, 'nchw')], Cout=64, stride=2),
    dict(B=1, H=13, W=19, src=[(3, 'nchw')], Cout=32, stride=2),
    dict(B=2, H=12, W=20, src=[(64, 'nhwc')], Cout=128, stride=2),
    dict(B=3, H=8, W=8, src=[(32, 'nhwc')], Cout=32, stride=1, residual=True),
    dict(B=2, H=8, W=12, src=[(32, 'nhwc'), (16, 'nhwc'), (2, 'nchw')], Cout=32,
         up=True),
    dict(B=1, H=4, W=4, src=[(512, 'nhwc'), (512, 'nhwc')], Cout=256, up=True),
    dict(B=2, H=9, W=7, src=[(20, 'nhwc')], Cout=48, stride=1, k=5, pad=2),
    dict(B=2, H=16, W=16, src=[(64, 'nhwc'), (64, 'nhwc'), (2, 'nchw')], Cout=32,
         up=True, act='mish'),
    # wide 3x3 stride-1 layers with enough tiles: Winograd.  F(2x2,3x3) (W % 4 != 0),
    # forward, data and weight gradient
    dict(B=4, H=8, W=18, src=[(256, 'nhwc')], Cout=320, stride=1, residual=True, wino=True),
    # F(4x4,3x3) forward / data gradient (64 tiles), F(2x2) weight gradient
    dict(B=8, H=8, W=16, src=[(320, 'nhwc')], Cout=256, stride=1, act='mish', wino=True),
    # fewer than 64 4x4 tiles: the 2x2 form on a 4-aligned image
    dict(B=6, H=8, W=16, src=[(256, 'nhwc')], Cout=384, stride=1, wino=True),
    # >= 128 4x4 tiles: the weight gradient takes the F(4x4,3x3) form too and
    # reuses the forward's transformed input
    dict(B=8, H=16, W=16, src=[(256, 'nhwc')], Cout=256, stride=1, wino=True),
    # large up-sampling layer with a flow member: four-lanes-per-pixel flow-gradient rows
    # and the matrix-core flat-member weight gradient
    dict(B=8, H=64, W=64, src=[(32, 'nhwc'), (32, 'nhwc'), (2, 'nchw')], Cout=64, up=True),
    # too few tiles: the direct kernel (transformed weights would dominate)
    dict(B=1, H=8, W=8, src=[(256, 'nhwc')], Cout=256, stride=1),
    # the first encoder layer's own kernels (csrc/first.hip: planar input, 64 outputs,
    # 8 x 32-pixel tiles, K = 9 C): ragged tiles in both directions, every column-block
    # count of the weight gradient (K + 1 = 28 .. 145 columns), Mish with its z copy
    dict(B=2, H=20, W=72, src=[(5, 'nchw')], Cout=64, stride=2, first=True),
    dict(B=1, H=16, W=64, src=[(12, 'nchw')], Cout=64, stride=2, act='mish', first=True),
    dict(B=3, H=34, W=18, src=[(9, 'nchw')], Cout=64, stride=2, first=True),
    dict(B=1, H=8, W=8, src=[(16, 'nchw')], Cout=64, stride=2, first=True),
    dict(B=1, H=8, W=8, src=[(3, 'nchw')], Cout=64, stride=2, first=True),
    dict(B=8, H=128, W=128, src=[(5, 'nchw')], Cout=64, stride=2, first=True),   # 256 tiles: one per group
    # decoder stages whose weight gradient takes the patch-resident f32 kernel
    # (csrc/wgrad_patch.hip): 64 input channels per workgroup (swapped halves of odd patch
    # slots), 144 blocks over 64 splits with a flat member beside the vector members
    dict(B=4, H=32, W=32, src=[(128, 'nhwc'), (128, 'nhwc')], Cout=64, up=True),
    dict(B=3, H=32, W=48, src=[(64, 'nhwc'), (192, 'nhwc'), (2, 'nchw')], Cout=64, up=True),
    # the finest decoder stage with its flow member folded away (two members of 64 -> 32):
    # forward by csrc/fwd_patch.hip (weights in registers, patch in LDS)
    dict(B=1, H=2, W=16, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
    dict(B=3, H=10, W=48, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True, act='mish'),
    dict(B=2, H=64, W=64, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
    # decoder stages whose exact-f32 forward is the nine-product form (csrc/fwd_min.hip: 4 | H,
    # 16 | W, two NHWC members of multiples of 32 channels): 4-row blocks with the K split over
    # the waves / 8-row blocks, members of different widths, Mish with its pre-activation copy,
    # blocks on every border of the frame, the coarsest benchmark stage's channel counts
    dict(B=2, H=12, W=32, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True, act='mish'),
    dict(B=1, H=8, W=16, src=[(32, 'nhwc'), (96, 'nhwc')], Cout=64, up=True),
    dict(B=3, H=24, W=48, src=[(64, 'nhwc'), (32, 'nhwc')], Cout=96, up=True, act='none'),
    dict(B=8, H=16, W=16, src=[(256, 'nhwc'), (256, 'nhwc')], Cout=128, up=True),
    dict(B=16, H=16, W=32, src=[(32, 'nhwc'), (32, 'nhwc')], Cout=32, up=True),
    # the coarsest and the finest decoder stage EXACTLY as benchmarked (batch 8, 256 x 256
    # input): 512 + 512 -> 256 at 16 x 16 (4-row blocks, K split over the waves, 32 chunks)
    # and 64 + 64 -> 32 at 128 x 128 (8-row blocks, 1 024 workgroups)
    dict(B=8, H=16, W=16, src=[(512, 'nhwc'), (512, 'nhwc')], Cout=256, up=True),
    dict(B=8, H=128, W=128, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
]


def build(case, seed=0):
    from dvs_of_training_framework_amd import conv as C
    g = torch.Generator().manual_seed(seed)
    B, H, W = case['B'], case['H'], case['W']
    k, stride = case.get('k', 3), case.get('stride', 1)
    pad, up = case.get('pad', 1), case.get('up', False)
    act = {'relu': C.ACT_RELU, 'mish': C.ACT_MISH, 'none': C.ACT_NONE}[
        case.get('act', 'relu')]
    xs = [torch.randn(B, c, H, W, generator=g) for c, _ in case['src']]
    ctot = sum(c for c, _ in case['src'])
    w = torch.randn(case['Cout'], ctot, k, k, generator=g) / (ctot * k * k) ** 0.5
    b = torch.randn(case['Cout'], generator=g)
    dev = [(x.cuda().contiguous() if lay == 'nchw' else nhwc(x))
           for x, (_, lay) in zip(xs, case['src'])]
    srcs = [(d, c, C.NCHW if lay == 'nchw' else C.NHWC)
            for d, (c, lay) in zip(dev, case['src'])]
    desc = C.make_desc(srcs, B, H, W, case['Cout'], k, stride, pad, up, act)
    desc._keepalive = dev   # the descriptor only holds raw pointers
    return C, xs, w, b, desc, act, dict(k=k, stride=stride, pad=pad, up=up)


def torch_fwd(xs, w, b, o, act, C, residual=None):
    inp = torch.cat(xs, 1)
    if o['up']:
        inp = F.interpolate(inp, scale_factor=2, mode='nearest')
    z = F.conv2d(inp, w, b, stride=o['stride'], padding=o['pad'])
    if residual is not None:
        z = z + residual
    y = F.relu(z) if act == C.ACT_RELU else F.mish(z) if act == C.ACT_MISH else z
    return y, z


# bf16 mode: operands rounded to bf16 (2^-9 relative) inside the MFMA kernels,
# f32 accumulation and storage; bound on the error relative to the peak
BF16_RTOL = 2e-2


# bf16x3: hi/lo split operands, three products: ~2^-16 per product
BF16X3_RTOL = 1e-4


@pytest.mark.parametrize('mfma', ['f32', 'bf16', 'bf16x3'])
@pytest.mark.parametrize('ci', range(len(CASES)))
def test_conv_fwd_dgrad_wgrad(ci, mfma, close=close):
    case = CASES[ci]
    C, xs, w, b, desc, act, o = build(case, seed=ci)
    if mfma != 'f32':
        desc.mfma = C.MFMA_BF16 if mfma == 'bf16' else C.MFMA_BF16X3
        exact, tol = close, BF16_RTOL if mfma == 'bf16' else BF16X3_RTOL

        def close(got, want):                      # noqa: F811
            exact(got, want, tol)
    xs = [x.requires_grad_(True) for x in xs]
    w.requires_grad_(True)
    b.requires_grad_(True)
    ho, wo = C.out_size(desc)
    res = torch.randn(case['B'], case['Cout'], ho, wo) if case.get('residual') else None
    y_ref, z_ref = torch_fwd(xs, w, b, o, act, C, res)
    import ctypes
    first = C._lib.lib().dvsof_conv2d_kernel_generation(ctypes.byref(desc), 0) == 3
    assert first == (bool(case.get('first')) or ci == 0)       # (case 0 has the first layer's shape too)
    nscratch = C._lib.lib().dvsof_conv2d_scratch_bytes(ctypes.byref(desc))
    # the Winograd path is the one under test (bf16-rounded operands stay direct)
    assert (nscratch > 0) == (bool(case.get('wino')) and mfma != 'bf16')
    w_dev = wphys(w.detach())
    w_fwd, wt = C.prepare(desc, w_dev, True)     # sub-pixel forms for up-layers
    # (Winograd layers: the weight gradient below reuses the forward's transformed
    # input when both run the same tile form -- the last case; the others recompute)
    y, z = C.conv_fwd(desc, w_fwd, b.cuda(), 'cuda',
                      nhwc(res) if res is not None else None, want_z=True,
                      keep_input_transform=bool(case.get('wino')))
    close(from_nhwc(y), y_ref)
    close(from_nhwc(z), z_ref)
    # backward w.r.t. the pre-activation output
    gz = torch.randn(z_ref.shape, generator=torch.Generator().manual_seed(99))
    z_ref.backward(gz)
    gz_d = nhwc(gz)
    ctot = sum(c for c, _ in case['src'])
    dsts, holders = [], []
    for x, (c, lay) in zip(xs, case['src']):
        buf = torch.empty(x.shape if lay == 'nchw' else
                          (x.shape[0], x.shape[2], x.shape[3], c), device='cuda')
        holders.append((buf, lay))
        dsts.append(dict(p=buf))
    C.conv_dgrad(desc, wt, gz_d, dsts)
    for (buf, lay), x in zip(holders, xs):
        close(buf if lay == 'nchw' else from_nhwc(buf), x.grad)
    dw = torch.empty(case['Cout'], o['k'], o['k'], ctot, device='cuda')
    db = torch.empty(case['Cout'], device='cuda')
    C.conv_wgrad(desc, gz_d, dw, db)
    close(dw.permute(0, 3, 1, 2), w.grad)
    close(db, b.grad)


@pytest.mark.parametrize('case', [
    dict(B=2, H=16, W=32, src=[(64, 'nhwc')], Cout=64, stride=1),
    dict(B=3, H=16, W=16, src=[(32, 'nhwc')], Cout=128, stride=2),        # odd number of 16-pixel groups per split
    dict(B=2, H=16, W=16, src=[(64, 'nhwc'), (32, 'nhwc'), (2, 'nchw')], Cout=32, up=True),   # sub-pixel phases + a flat member
    dict(B=8, H=64, W=64, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),                # 32 x 128 tile, many K splits
    dict(B=4, H=16, W=16, src=[(256, 'nhwc')], Cout=256, stride=1),                          # direct wide layer (no Winograd in mode 3)
    # the patch-resident decoder kernel (csrc/wgrad_patch.hip; the 64 x 64 case above too):
    # 16-pixel rows (every group is its own row: top / bottom / left / right borders in one
    # group), two output-channel tiles, unequal members, an odd number of groups per split
    dict(B=3, H=16, W=16, src=[(64, 'nhwc'), (32, 'nhwc')], Cout=64, up=True),
    dict(B=2, H=8, W=32, src=[(32, 'nhwc')], Cout=32, up=True),
    dict(B=1, H=16, W=48, src=[(96, 'nhwc'), (32, 'nhwc')], Cout=96, up=True),
    # 64 input channels per workgroup (members of 64 | C and >= 512 workgroups): 128-byte
    # pixel slots with the swizzled halves; 144 blocks over 64 splits (empty splits write zeros)
    dict(B=4, H=32, W=32, src=[(128, 'nhwc'), (128, 'nhwc')], Cout=64, up=True),
    dict(B=3, H=32, W=48, src=[(64, 'nhwc'), (192, 'nhwc'), (2, 'nchw')], Cout=64, up=True),
    # the finest decoder stage with its flow member folded away (two members of 64 -> 32):
    # forward by csrc/fwd_patch.hip (weights in registers, patch in LDS)
    dict(B=1, H=2, W=16, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
    dict(B=3, H=10, W=48, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True, act='mish'),
    dict(B=2, H=64, W=64, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
    # decoder stages whose exact-f32 forward is the nine-product form (csrc/fwd_min.hip: 4 | H,
    # 16 | W, two NHWC members of multiples of 32 channels): 4-row blocks with the K split over
    # the waves / 8-row blocks, members of different widths, Mish with its pre-activation copy,
    # blocks on every border of the frame, the coarsest benchmark stage's channel counts
    dict(B=2, H=12, W=32, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True, act='mish'),
    dict(B=1, H=8, W=16, src=[(32, 'nhwc'), (96, 'nhwc')], Cout=64, up=True),
    dict(B=3, H=24, W=48, src=[(64, 'nhwc'), (32, 'nhwc')], Cout=96, up=True, act='none'),
    dict(B=8, H=16, W=16, src=[(256, 'nhwc'), (256, 'nhwc')], Cout=128, up=True),
    dict(B=16, H=16, W=32, src=[(32, 'nhwc'), (32, 'nhwc')], Cout=32, up=True),
    # the coarsest and the finest decoder stage EXACTLY as benchmarked (batch 8, 256 x 256
    # input): 512 + 512 -> 256 at 16 x 16 (4-row blocks, K split over the waves, 32 chunks)
    # and 64 + 64 -> 32 at 128 x 128 (8-row blocks, 1 024 workgroups)
    dict(B=8, H=16, W=16, src=[(512, 'nhwc'), (512, 'nhwc')], Cout=256, up=True),
    dict(B=8, H=128, W=128, src=[(64, 'nhwc'), (64, 'nhwc')], Cout=32, up=True),
])
def test_wgrad_on_bf16_twins_equals_the_operand_mode(case):
    """mfma mode 3: the vector members' weight gradient streams the bf16 TWINS
    of gout and of the sources through LDS (ds_read_b64_tr_b16 transposed
    fragment reads, K = pixels).  Mode 1 rounds the same f32 values to bf16 in
    registers, so both multiply identical operands: equal up to f32 summation
    order.  Bias gradient: sums of the bf16-rounded gout (2^-9 per element)."""
    C, xs, w, b, desc, act, o = build(case, seed=11)
    ho, wo = C.out_size(desc)
    g = torch.Generator().manual_seed(5)
    gz = torch.randn(case['B'], case['Cout'], ho, wo, generator=g)
    gz_d = nhwc(gz)
    ctot = sum(c for c, _ in case['src'])

    def run(mode, twins):
        srcs = []
        for t, (c, lay) in zip(desc._keepalive, case['src']):
            t16 = t.to(torch.bfloat16) if (twins and lay == 'nhwc') else None
            srcs.append((t, c, C.NCHW if lay == 'nchw' else C.NHWC, t16))
        d = C.make_desc(srcs, case['B'], case['H'], case['W'], case['Cout'], o['k'], o['stride'],
                        o['pad'], o['up'], act, mode)
        d._keep = srcs
        dw = torch.empty(case['Cout'], o['k'], o['k'], ctot, device='cuda')
        db = torch.empty(case['Cout'], device='cuda')
        C.conv_wgrad(d, gz_d, dw, db, gz_d.to(torch.bfloat16) if twins else None)
        torch.cuda.synchronize()
        return dw, db
    dw1, db1 = run(C.MFMA_BF16, False)
    dw3, db3 = run(C.MFMA_BF16_TWINS, True)
    close(dw3, dw1, 1e-5)
    close(db3, db1, 1e-2)
    dw3b, _ = run(C.MFMA_BF16_TWINS, True)
    assert torch.equal(dw3, dw3b)                 # fixed-order reductions
    # and both stay within the bf16 bound of the exact gradient
    dwx, _ = run(C.MFMA_F32, False)
    close(dw3, dwx, BF16_RTOL)


@pytest.mark.parametrize('B,H,W,act,cls,want_z', [
    (1, 2, 16, 'relu', False, False),       # one block: every patch row / column border at once
    (2, 8, 16, 'relu', True, True),         # border-class bias of a folded flow member, z copy
    (3, 10, 48, 'mish', True, True),        # strips shared between workgroups, ragged last range
    (8, 64, 64, 'relu', True, False),       # 512 persistent workgroups x 4 blocks
    (8, 128, 128, 'relu', False, False),    # the benchmark's finest stage: 8 blocks per workgroup
])
def test_finest_decoder_forward_on_bf16_twins_equals_the_operand_mode(B, H, W, act, cls, want_z):
    """csrc/fwd_patch.hip (mode 3, cat[x 64, skip 64] -> 32, sub-pixel phases):
    input patch resident in LDS, weights in registers.  Mode 1 rounds the same
    f32 values to bf16 in registers, so both multiply identical operands: equal
    up to f32 summation order; the bf16 twin of y is y rounded; and both stay
    within the bf16 bound of the exact layer."""
    from dvs_of_training_framework_amd import conv as C
    g = torch.Generator().manual_seed(B * 131 + H)
    x = nhwc(torch.randn(B, 64, H, W, generator=g))
    sk = nhwc(torch.randn(B, 64, H, W, generator=g))
    w = wphys(torch.randn(32, 128, 3, 3, generator=g) / (128 * 9) ** 0.5)
    b = torch.randn(32, generator=g).cuda()
    b_cls = (torch.randn(9, 32, generator=g).cuda() * 0.3) if cls else None
    a = {'relu': C.ACT_RELU, 'mish': C.ACT_MISH}[act]

    def run(mode):
        twins = mode == C.MFMA_BF16_TWINS
        srcs = [(t, 64, C.NHWC, t.to(torch.bfloat16) if twins else None) for t in (x, sk)]
        d = C.make_desc(srcs, B, H, W, 32, 3, 1, 1, True, a, mode)
        d._keep = srcs
        if twins:
            w_f, _, w_f16, _ = C.prepare(d, w, False, want16=True)
        else:
            (w_f, _), w_f16 = C.prepare(d, w, False), None
        y, z = C.conv_fwd(d, w_f, b, 'cuda', None, want_z=want_z, weight16=w_f16, bias_cls=b_cls)
        torch.cuda.synchronize()
        return y, z, d._y16
    y1, z1, _ = run(C.MFMA_BF16)
    y3, z3, y16 = run(C.MFMA_BF16_TWINS)
    close(y3, y1, 1e-5)
    if want_z:
        close(z3, z1, 1e-5)
    assert torch.equal(y16.view(y3.shape), y3.to(torch.bfloat16))
    y3b, _, _ = run(C.MFMA_BF16_TWINS)
    assert torch.equal(y3, y3b)
    yx, _, _ = run(C.MFMA_F32)
    close(y3, yx, BF16_RTOL)


@pytest.mark.parametrize('B,H,W,Cx,Cs,Cout', [
    (2, 8, 8, 32, 32, 32),
    (3, 6, 10, 64, 32, 64),          # odd frame sides, unequal members
    (8, 32, 32, 64, 64, 32),         # a decoder-stage shape (K splits, sub-pixel fold)
])
def test_flow_member_folded_into_weight_space(B, H, W, Cx, Cs, Cout):
    """Backward of a decoder stage whose input is cat[x, skip, flow] with
    flow = Wh x + bh (the previous stage's head): the data gradient on
    cat[x, skip] with the folded weights, the weight gradient with the flow
    columns derived from the x columns + border sums, and the head's gradient
    additions (csrc/flowfold.hip) against ATen autograd through
    cat / interpolate / conv2d."""
    from dvs_of_training_framework_amd import conv as C
    g = torch.Generator().manual_seed(B * 7 + Cout)
    x = torch.randn(B, Cx, H, W, generator=g, requires_grad=True)
    sk = torch.randn(B, Cs, H, W, generator=g, requires_grad=True)
    wh = (torch.randn(2, Cx, generator=g) / Cx ** 0.5).requires_grad_(True)
    bh = torch.randn(2, generator=g, requires_grad=True)
    ctot = Cx + Cs + 2
    w = (torch.randn(Cout, ctot, 3, 3, generator=g) / (ctot * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    flow = F.conv2d(x, wh[:, :, None, None], bh)
    inp = F.interpolate(torch.cat([x, sk, flow], 1), scale_factor=2, mode='nearest')
    z = F.conv2d(inp, w, b, padding=1)
    gz = torch.randn(z.shape, generator=g)
    z.backward(gz)

    x_d, sk_d = nhwc(x.detach()), nhwc(sk.detach())
    flow_d = flow.detach().cuda().contiguous()
    w_d, wh_d, bh_d = wphys(w.detach()), wh.detach().cuda().contiguous(), bh.detach().cuda()
    gz_d = nhwc(gz)
    # data gradient: two vector members, folded weights
    d2 = C.make_desc([(x_d, Cx, C.NHWC), (sk_d, Cs, C.NHWC)], B, H, W, Cout, 3, 1, 1, True)
    w_eff = C.flow_fold_weights(w_d, Cout, ctot, 0, Cx, Cx + Cs, wh_d)
    w_f, w_dg = C.prepare(d2, w_eff, True)
    # forward on cat[x, skip]: the head's bias arrives through the in-frame flow
    # taps only -> a bias per border class of the output pixel
    b_eff, b_cls = C.flow_fold_bias(w_d, Cout, ctot, Cx + Cs, bh_d, b.detach().cuda())
    y_f, z_f = C.conv_fwd(d2, w_f, b_eff, 'cuda', None, want_z=True, bias_cls=b_cls)
    close(from_nhwc(z_f), z)
    d3f = C.make_desc([(x_d, Cx, C.NHWC), (sk_d, Cs, C.NHWC), (flow_d, 2, C.NCHW)], B, H, W, Cout,
                      3, 1, 1, True)
    w_f3, _ = C.prepare(d3f, w_d, False)
    _, z_3 = C.conv_fwd(d3f, w_f3, b.detach().cuda(), 'cuda', None, want_z=True)
    close(from_nhwc(z_f), from_nhwc(z_3), 1e-5)
    gx = torch.empty(B, H, W, Cx, device='cuda')
    gs = torch.empty(B, H, W, Cs, device='cuda')
    C.conv_dgrad(d2, w_dg, gz_d, [dict(p=gx), dict(p=gs)])
    close(from_nhwc(gx), x.grad)          # includes the path through the flow head
    close(from_nhwc(gs), sk.grad)
    # weight gradient: vector members on the matrix cores, flow columns in weight space
    d3 = C.make_desc([(x_d, Cx, C.NHWC), (sk_d, Cs, C.NHWC), (flow_d, 2, C.NCHW)], B, H, W, Cout,
                     3, 1, 1, True)
    dw = torch.full((Cout, 3, 3, ctot), float('nan'), device='cuda')
    db = torch.empty(Cout, device='cuda')
    C.conv_wgrad(d3, gz_d, dw, db, skip_flat=True)
    dwh = torch.zeros(2, Cx, device='cuda')
    dbh = torch.zeros(2, device='cuda')
    C.flow_fold_grads(dw, w_d, Cout, ctot, 0, Cx, Cx + Cs, wh_d, bh_d, db, gz_d, B, 2 * H, 2 * W,
                      dwh, dbh)
    close(dw.permute(0, 3, 1, 2), w.grad)
    close(db, b.grad)
    close(dwh, wh.grad)
    close(dbh, bh.grad)
    # the unfused path (flat-member kernels) gives the same weight gradient
    dw2 = torch.empty(Cout, 3, 3, ctot, device='cuda')
    C.conv_wgrad(d3, gz_d, dw2, db)
    close(dw, dw2, 1e-5)


@pytest.mark.parametrize('B,H,W,Cin,Cout,act', [
    (2, 8, 12, 32, 64, 'relu'),
    (1, 5, 7, 64, 48, 'mish'),        # odd sizes, Cout not a tile multiple
    (8, 32, 32, 128, 64, 'relu'),     # a decoder-stage shape
])
def test_transposed_conv_layer(B, H, W, Cin, Cout, act):
    """upsample = UP_ZERO: 2x zero insertion + 3x3/pad 1 = transposed
    convolution with stride 2 (north star: "strided conv / transposed-conv /
    Mish stack"), run as four output-parity phases on the matrix cores.
    Forward, data gradient (a stride-2 convolution of the output gradient) and
    weight / bias gradient (the adjoint stride-2 layer's, flip-transposed)
    against ATen on the zero-inserted input and against conv_transpose2d."""
    from dvs_of_training_framework_amd import conv as C
    g = torch.Generator().manual_seed(B * 100 + H)
    a = {'relu': C.ACT_RELU, 'mish': C.ACT_MISH}[act]
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    xw = torch.stack([x, torch.zeros_like(x)], dim=-1).reshape(B, Cin, H, 2 * W)
    xz = torch.stack([xw, torch.zeros_like(xw)], dim=-2).reshape(B, Cin, 2 * H, 2 * W)
    assert torch.equal(xz[:, :, ::2, ::2], x) and int((xz != 0).sum()) == int((x != 0).sum())
    z_ref = F.conv2d(xz, w, b, padding=1)
    z_t = F.conv_transpose2d(x, w.flip(2, 3).transpose(0, 1), b, stride=2, padding=1,
                             output_padding=1)
    close(z_t, z_ref, 1e-5)       # the layer IS conv_transpose2d (flipped kernel)
    y_ref = F.relu(z_ref) if act == 'relu' else F.mish(z_ref)
    x_d = nhwc(x.detach())
    desc = C.make_desc([(x_d, Cin, C.NHWC)], B, H, W, Cout, 3, 1, 1, C.UP_ZERO, a)
    assert C.out_size(desc) == (2 * H, 2 * W)
    w_fwd, w_dg = C.prepare(desc, wphys(w.detach()), True)
    assert w_fwd.numel() == 16 * Cout * Cin and w_dg.numel() == 9 * Cout * Cin
    y, z = C.conv_fwd(desc, w_fwd, b.detach().cuda(), 'cuda', None, want_z=True)
    close(from_nhwc(z), z_ref)
    close(from_nhwc(y), y_ref)
    gz = torch.randn(z_ref.shape, generator=g)
    z_ref.backward(gz)
    gz_d = nhwc(gz)
    gx = torch.empty(B, H, W, Cin, device='cuda')
    C.conv_dgrad(desc, w_dg, gz_d, [dict(p=gx)])
    close(from_nhwc(gx), x.grad)
    dw = torch.empty(Cout, 3, 3, Cin, device='cuda')
    db = torch.empty(Cout, device='cuda')
    C.conv_wgrad(desc, gz_d, dw, db)
    close(dw.permute(0, 3, 1, 2), w.grad)
    close(db, b.grad)


def test_transposed_conv_rejects_what_it_does_not_implement():
    from dvs_of_training_framework_amd import conv as C
    x = torch.zeros(1, 4, 4, 32, device='cuda')
    for bad in (dict(ksize=5, pad=2), dict(stride=2)):
        kw = dict(ksize=3, stride=1, pad=1)
        kw.update(bad)
        d = C.make_desc([(x, 32, C.NHWC)], 1, 4, 4, 32, upsample=C.UP_ZERO, **kw)
        with pytest.raises(RuntimeError):
            C.conv_fwd(d, torch.zeros(32 * 16 * 32, device='cuda'), None, 'cuda')
    x5 = torch.zeros(1, 5, 4, 4, device='cuda')      # planar 5-channel source
    d = C.make_desc([(x5, 5, C.NCHW)], 1, 4, 4, 32, upsample=C.UP_ZERO)
    with pytest.raises(RuntimeError):
        C.conv_fwd(d, torch.zeros(32 * 16 * 5, device='cuda'), None, 'cuda')


@pytest.mark.parametrize('case', [
    dict(B=2, H=8, W=8, src=[(32, 'nhwc')], Cout=64, stride=2),
    dict(B=8, H=16, W=8, src=[(256, 'nhwc')], Cout=256, stride=1),     # Winograd F(4x4,3x3)
])
def test_dgrad_epilogue_addends_and_act(case):
    from dvs_of_training_framework_amd import conv as C
    C, xs, w, b, desc, act, o = build(case, seed=3)
    x = xs[0].requires_grad_(True)
    _, z_ref = torch_fwd([x], w, b, o, act, C)
    gz = torch.randn(z_ref.shape)
    z_ref.backward(gz)
    a1, a2 = torch.randn(x.shape), torch.randn(x.shape)
    ysrc = torch.randn(x.shape)
    want = (x.grad + a1 + a2) * (ysrc > 0).float()
    buf = torch.empty(case['B'], case['H'], case['W'], case['src'][0][0], device='cuda')
    _, wt = C.prepare(desc, wphys(w), True)
    C.conv_dgrad(desc, wt, nhwc(gz), [dict(p=buf, addend=nhwc(a1), addend2=nhwc(a2),
                                           actsrc=nhwc(ysrc))], C.ACT_RELU)
    close(from_nhwc(buf), want)


def test_decoder_dgrad_nine_product_form_epilogue():
    """csrc/dgrad_min.hip (two members of 64 | C, 8 | H): the gradient lands in
    the member its channel tile belongs to; addends and act' on one member only
    (the decoder's first stage: x = the last residual output)."""
    from dvs_of_training_framework_amd import conv as C
    case = dict(B=2, H=16, W=32, src=[(128, 'nhwc'), (64, 'nhwc')], Cout=48 + 16, up=True)
    C, xs, w, b, desc, act, o = build(case, seed=5)
    xs = [x.requires_grad_(True) for x in xs]
    _, z_ref = torch_fwd(xs, w, b, o, act, C)
    gz = torch.randn(z_ref.shape)
    z_ref.backward(gz)
    a1, a2, ysrc = (torch.randn(xs[0].shape) for _ in range(3))
    want0 = (xs[0].grad + a1 + a2) * (ysrc > 0).float()
    bufs = [torch.empty(2, 16, 32, c, device='cuda') for c, _ in case['src']]
    _, wt = C.prepare(desc, wphys(w), True)
    C.conv_dgrad(desc, wt, nhwc(gz), [dict(p=bufs[0], addend=nhwc(a1), addend2=nhwc(a2),
                                           actsrc=nhwc(ysrc)), dict(p=bufs[1])], C.ACT_RELU)
    close(from_nhwc(bufs[0]), want0)
    close(from_nhwc(bufs[1]), xs[1].grad)
    # a flow head on member 0 folded into the epilogue (dvsof_grad_dst_t.head_w):
    # + W_h^T g_flow ahead of act'; the head's own weight gradient with gx=None
    assert C.dgrad_fuses_head(desc)
    wh, gf = torch.randn(2, 128) / 8, torch.randn(2, 2, 16, 32)
    want_h = (xs[0].grad + a1 + torch.einsum('kc,bkyx->bcyx', wh, gf)) * (ysrc > 0).float()
    C.conv_dgrad(desc, wt, nhwc(gz), [dict(p=bufs[0], addend=nhwc(a1), actsrc=nhwc(ysrc),
                                           head_w=wh.cuda(), head_gflow=gf.cuda()),
                                      dict(p=bufs[1])], C.ACT_RELU)
    close(from_nhwc(bufs[0]), want_h)
    close(from_nhwc(bufs[1]), xs[1].grad)
    dw, db = torch.empty(2, 128, device='cuda'), torch.empty(2, device='cuda')
    C.head_bwd(nhwc(ysrc), wh.cuda(), gf.cuda(), None, None, C.ACT_RELU, None, dw, db,
               2, 16, 32, 128)
    close(dw, torch.einsum('bkyx,bcyx->kc', gf, ysrc))
    close(db, gf.sum((0, 2, 3)))
    # ... or from per-block partials the same epilogue leaves (head_x / head_part), with the
    # head's input apart from actsrc (Mish) and the same tensor (ReLU)
    for hx in (torch.randn(ysrc.shape), ysrc):
        part = C.dgrad_head_part(desc, 128, 'cuda')
        assert part is not None and part.shape == (2 * (16 // 8) * (32 // 16), 2 * 128 + 2)
        part.fill_(float('nan'))        # every element is written
        hx_d = nhwc(hx)
        as_d = nhwc(ysrc) if hx is not ysrc else hx_d
        C.conv_dgrad(desc, wt, nhwc(gz), [dict(p=bufs[0], addend=nhwc(a1), actsrc=as_d,
                                               head_w=wh.cuda(), head_gflow=gf.cuda(),
                                               head_x=hx_d, head_part=part),
                                          dict(p=bufs[1])], C.ACT_RELU)
        close(from_nhwc(bufs[0]), want_h)
        dw2, db2 = torch.empty(2, 128, device='cuda'), torch.empty(2, device='cuda')
        C.head_reduce(part, 128, dw2, db2)
        close(dw2, torch.einsum('bkyx,bcyx->kc', gf, hx))
        close(db2, gf.sum((0, 2, 3)))
    # the same fold in the general kernels' epilogue (here: bf16 operands, where the layer's data
    # gradient is the 4x4 stride-2 form on gconv2) -- and with a third, planar member beside it
    desc.mfma = C.MFMA_BF16
    assert C.dgrad_fuses_head(desc)
    _, wt16 = C.prepare(desc, wphys(w), True)
    C.conv_dgrad(desc, wt16, nhwc(gz), [dict(p=bufs[0], addend=nhwc(a1), actsrc=nhwc(ysrc),
                                             head_w=wh.cuda(), head_gflow=gf.cuda()),
                                        dict(p=bufs[1])], C.ACT_RELU)
    close(from_nhwc(bufs[0]), want_h, BF16_RTOL)
    close(from_nhwc(bufs[1]), xs[1].grad, BF16_RTOL)
    desc.mfma = C.MFMA_F32
    # ... and refused where the kernel is another one (here: stride 1, no up-sampling)
    case2 = dict(B=1, H=8, W=16, src=[(64, 'nhwc')], Cout=64, up=False)
    C2, xs2, w2, b2, desc2, act2, o2 = build(case2, seed=6)
    assert not C.dgrad_fuses_head(desc2)
    _, wt2 = C.prepare(desc2, wphys(w2), True)
    with pytest.raises(Exception):
        C.conv_dgrad(desc2, wt2, torch.zeros(1, 8, 16, 64, device='cuda'),
                     [dict(p=torch.empty(1, 8, 16, 64, device='cuda'), head_w=wh.cuda(),
                           head_gflow=gf.cuda())], C.ACT_RELU)


@pytest.mark.parametrize('B,H,W', [(8, 16, 16), (4, 8, 32), (2, 24, 32)])
def test_winograd_chain_forms_are_the_consumers_own(B, H, W):
    """dvsof_conv_desc_t.winograd_next / winograd_next_gout / winograd_pre /
    winograd_gout (csrc/winograd.hip wino_chain_kernel): a Winograd layer's
    output transform also writes its consumer's transformed input and gradient
    form.  Bitwise the same outputs, gradients and weight gradients as the
    three separate transforms -- forward with residual and Mish (pre-activation
    copy), data gradient with addend and act'."""
    from dvs_of_training_framework_amd import conv as C
    Cc = 256
    case = dict(B=B, H=H, W=W, src=[(Cc, 'nhwc')], Cout=Cc, stride=1, act='mish')
    C, xs, w1, b1, d1, act, o = build(case, seed=3)
    _, _, w2, b2, d2, _, _ = build(case, seed=4)
    assert C.winograd_tile(d1, 0) == 4 and C.winograd_chain(d1, 0) and C.winograd_chain(d1, 1)
    dev = 'cuda'
    u1, u1t = C.prepare(d1, wphys(w1), True)
    u2, u2t = C.prepare(d2, wphys(w2), True)
    res = nhwc(torch.randn(B, Cc, H, W))

    def layer2_src(y):      # layer 2 reads layer 1's output
        d2.src[0].p = y.data_ptr()
        return y
    # ---- plain: every call makes its own transforms
    y1, z1 = C.conv_fwd(d1, u1, b1.cuda(), dev, residual=res, want_z=True, keep_input_transform=True)
    layer2_src(y1)
    y2, _ = C.conv_fwd(d2, u2, b2.cuda(), dev, keep_input_transform=True)
    g2 = nhwc(torch.randn(B, Cc, H, W))
    add = nhwc(torch.randn(B, Cc, H, W))
    g1 = torch.empty_like(y1)
    C.conv_dgrad(d2, u2t, g2, [dict(p=g1, addend=add, actsrc=z1)], act)
    dw1, db1 = torch.empty(Cc * 9 * Cc, device=dev), torch.empty(Cc, device=dev)
    C.conv_wgrad(d1, g1, dw1, db1)
    g0 = torch.empty_like(y1)
    C.conv_dgrad(d1, u1t, g1, [dict(p=g0)], act)
    # ---- chained
    v2 = C.winograd_form(d1, Cc, dev)
    y1c, z1c = C.conv_fwd(d1, u1, b1.cuda(), dev, residual=res, want_z=True,
                          keep_input_transform=True, wino_next=v2)
    layer2_src(y1c)
    y2c, _ = C.conv_fwd(d2, u2, b2.cuda(), dev, keep_input_transform=True, wino_pre=v2)
    vg, zg = C.winograd_form(d2, Cc, dev), C.winograd_form(d2, Cc, dev)
    g1c = torch.empty_like(y1)
    C.conv_dgrad(d2, u2t, g2, [dict(p=g1c, addend=add, actsrc=z1c)], act,
                 wino_next=vg, wino_next_gout=zg)
    dw1c, db1c = torch.empty_like(dw1), torch.empty_like(db1)
    C.conv_wgrad(d1, g1c, dw1c, db1c, wino_gout=zg if C.winograd_tile(d1, 2) == 4 else None)
    g0c = torch.empty_like(y1)
    C.conv_dgrad(d1, u1t, g1c, [dict(p=g0c)], act, wino_pre=vg)
    for a, b_ in ((y1, y1c), (z1, z1c), (y2, y2c), (g1, g1c), (dw1, dw1c), (db1, db1c), (g0, g0c)):
        assert torch.equal(a, b_)
    # a layer that is not a Winograd evaluation refuses the options
    case3 = dict(B=1, H=8, W=16, src=[(64, 'nhwc')], Cout=64, up=False)
    C3, xs3, w3, b3, d3, act3, o3 = build(case3, seed=6)
    assert not C.winograd_chain(d3, 0)
    with pytest.raises(Exception):
        C.conv_fwd(d3, wphys(w3), b3.cuda(), dev, wino_next=v2)


@pytest.mark.parametrize('Cc,act', [(32, 'relu'), (256, 'relu'), (64, 'mish')])
def test_flow_head(Cc, act):
    from dvs_of_training_framework_amd import conv as C
    B, H, W = 2, 12, 20
    aid = C.ACT_RELU if act == 'relu' else C.ACT_MISH
    x = torch.randn(B, Cc, H, W, requires_grad=True)
    w = (torch.randn(2, Cc, 1, 1) / Cc ** 0.5).requires_grad_(True)
    b = torch.randn(2, requires_grad=True)
    zsrc = torch.randn(B, Cc, H, W)      # stand-in for the producer's y / z
    f_ref = F.conv2d(x, w, b)
    f = C.head_fwd(nhwc(x), w.view(2, Cc).cuda().contiguous(), b.cuda(), B, H, W, Cc)
    close(f, f_ref)
    gf, gx_in = torch.randn(f_ref.shape), torch.randn(x.shape)
    f_ref.backward(gf)
    dact = (zsrc > 0).float() if act == 'relu' else \
        torch.autograd.grad(F.mish(zsrc.requires_grad_(True)).sum(), zsrc)[0]
    want_gx = (x.grad + gx_in) * dact
    gx = torch.empty(B, H, W, Cc, device='cuda')
    dw, db = torch.empty(2, Cc, device='cuda'), torch.empty(2, device='cuda')
    C.head_bwd(nhwc(x), w.view(2, Cc).cuda().contiguous(), gf.cuda(), nhwc(gx_in),
               nhwc(zsrc.detach()), aid, gx, dw, db, B, H, W, Cc)
    close(from_nhwc(gx), want_gx)
    close(dw, w.grad.view(2, Cc))
    close(db, b.grad)


@pytest.mark.parametrize('mish,shape', [(False, (2, 5, 32, 48)), (True, (2, 5, 32, 48)),
                                        # residual stage 8 x 8 x 8: 128 2x2 tiles -> Winograd
                                        (False, (8, 5, 128, 128))])
def test_predictor_vs_torch_reference(mish, shape):
    """Whole predictor forward + backward (explicit schedule) vs ATen autograd."""
    from dvs_of_training_framework_amd.predictor import Predictor
    from oracle.ref_model import ref_predictor
    torch.manual_seed(1)
    B, Cin, H, W = shape
    act = torch.nn.Mish() if mish else torch.nn.ReLU()
    net = Predictor(Cin, act)
    x = torch.randn(B, Cin, H, W)
    state = {k: v.detach().clone().requires_grad_(True)
             for k, v in net.state_dict().items()}
    ref_flows = ref_predictor(state, x, mish)
    gfl = [torch.randn(f.shape) for f in ref_flows]
    sum((f * g).sum() for f, g in zip(ref_flows, gfl)).backward()
    net = net.cuda()
    flows = net(x.cuda())
    assert [tuple(f.shape) for f in flows] == [(B, 2, H // 8, W // 8), (B, 2, H // 4, W // 4),
                                               (B, 2, H // 2, W // 2), (B, 2, H, W)]
    for f, r in zip(flows, ref_flows):
        close(f, r, 1e-3)
    sum((f * g.cuda()).sum() for f, g in zip(flows, gfl)).backward()
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        close(p.grad, state[name].grad, 1e-3)
