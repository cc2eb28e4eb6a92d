"""EV_FlowNet predictor (docs/MODEL_SPEC.md) on the HIP conv stack.

The reference imports ``EV_FlowNet.net.Model`` whose source is an un-vendored
submodule; what the reference pins is the contract:
4 flow maps coarse to fine at ``imsize // 2**i`` (DummyNet/net.py:59-66,
tests/training/test_training.py:45-46), attribute ``predictor`` with
state-dict names ``predictor.enc.N.conv.*`` (train_flownet.py:50-54,79-85).
The architecture is the canonical EV-FlowNet (Zhu et al., RSS 2018):

  enc.i   3x3 stride-2 conv + act,  C -> 64 -> 128 -> 256 -> 512
  res.i   x + conv2(act(conv1(x))), then act          (2 blocks at 512)
  dec.i   3x3 conv + act on the 2x nearest-upsampled concat[x, skip, flow]
          -> 256, 128, 64, 32;  dec.i.flow = 1x1 conv -> 2 (linear head)

Forward and backward are ONE autograd node with an explicit schedule: the
backward pass is ~45 launches of the gather-conv / wgrad / head kernels with
the gradient sums and activation derivatives folded into kernel epilogues
(no autograd graph, no cat/upsample tensors, no elementwise passes).
"""
import contextlib
import math
import os

import torch
from torch import nn

from . import conv as C

_SIDE_STREAMS = {}      # device index -> the second backward stream
_EXTRA_STREAMS = {}     # device index -> further streams (DVSOF_WGRAD_STREAMS > 1)

ENC_CH = (64, 128, 256, 512)
DEC_CH = (256, 128, 64, 32)
NUM_RES = 2


def activation_id(activation):
    """nn.ReLU -> ACT_RELU, Mish-like module -> ACT_MISH
    (reference: utils/options.py:341-347 passes nn.ReLU() or Mish())."""
    if activation is None or isinstance(activation, nn.ReLU):
        return C.ACT_RELU
    if isinstance(activation, nn.Identity):
        return C.ACT_NONE
    if type(activation).__name__.lower() == 'mish':
        return C.ACT_MISH
    raise ValueError(f'unsupported activation {activation!r}: the HIP conv '
                     'stack fuses ReLU or Mish')


class ConvParams(nn.Module):
    """weight [Cout,Cin,k,k] stored channels_last = physical [Cout][k][k][Cin]
    (the K-contiguous layout the MFMA kernels stream), bias [Cout]."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        w = torch.empty(cout, cin, k, k).contiguous(
            memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(cout))
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv2d's default init
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(self.cin * self.k * self.k)
        nn.init.uniform_(self.bias, -bound, bound)


class _Named(nn.Module):
    def __init__(self, **mods):
        super().__init__()
        for k, v in mods.items():
            self.add_module(k, v)


def _prep(desc, weight, want_dgrad, phase_weights=None, want16=False):
    """conv.prepare -> always (w_fwd, w_dgrad, w_fwd16, w_dgrad16)."""
    out = C.prepare(desc, weight, want_dgrad, phase_weights, want16)
    return out if want16 else (out[0], out[1], None, None)


def _phys(w):
    """Physical [Cout][k][k][Cin] buffer of a channels_last OIHW weight."""
    assert w.permute(0, 2, 3, 1).is_contiguous(), \
        'conv weights must stay in channels_last memory format'
    return w


class _PredictorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, module, want_grad, *params):
        # params: enc(w,b)x4, res(w1,b1,w2,b2)x2, dec(w,b,fw,fb)x4
        dev = x0.device
        act = module.act
        B, Cin, H, W = x0.shape
        mish = act == C.ACT_MISH
        twins = module.mfma == C.MFMA_BF16_TWINS
        p = list(params)
        enc = [(p[2 * i], p[2 * i + 1]) for i in range(4)]
        o = 8
        res = [(p[o + 4 * i], p[o + 4 * i + 1], p[o + 4 * i + 2],
                p[o + 4 * i + 3]) for i in range(NUM_RES)]
        o += 4 * NUM_RES
        dec = [(p[o + 4 * i], p[o + 4 * i + 1], p[o + 4 * i + 2],
                p[o + 4 * i + 3]) for i in range(4)]
        L = []          # per conv layer: dict(desc, y, z, srcs)
        # data-gradient forms of the weights are only needed by the backward:
        # they are made on the second stream while the forward runs.  That
        # stream first waits for everything enqueued so far (the optimizer's
        # update of the weights; the previous backward's readers of the
        # buffers it is about to reuse).
        main = torch.cuda.current_stream(dev)
        side = module._wgrad_stream(dev) if want_grad else None
        begin = module._take_step_begin()
        if side is not None:
            # ... or, when the caller marked the start of the step (mark_step_begin, ahead
            # of the voxeliser), only for what preceded THAT: nothing made here reads the
            # event volume, so the forms run beside the voxeliser and the first layers
            if begin is not None:
                side.wait_event(begin)
            else:
                side.wait_stream(main)

        # bf16-twins mode: the twins of the weights that are used as they are
        # (stride-2 encoder layers, direct residual layers) in ONE launch; the
        # twins of prepared forms come from the kernels that make the forms
        raw16, raw16_ready = {}, None
        if twins:
            raws = [e_[0] for e_ in enc] + [r_[j] for r_ in res for j in (0, 2)]
            # on the second stream when there is one (16 us off the forward's lane): the
            # first layer does not read them, the main stream waits before the second
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                raw16 = {id(w_): t_ for w_, t_ in
                         zip(raws, C.to_bf16_many([_phys(w_) for w_ in raws]))}
                if side is not None:
                    raw16_ready = torch.cuda.Event()
                    raw16_ready.record(side)

        # Prepared FORWARD forms (Winograd-domain weights of the residual
        # layers, sub-pixel phase kernels of the decoder) are not needed
        # before the encoder has run: when training they are made on the
        # second stream too, ahead of the data-gradient forms, and the main
        # stream waits for them once, at the first residual layer.
        pre, pre_ready = {}, None

        def _fold_pre(srcs2, hh, ww, cout_, wgt_, bias_, head):
            cx, cs = srcs2[0][1], srcs2[1][1]
            ctot = cx + cs + 2
            w_eff = C.flow_fold_weights(_phys(wgt_), cout_, ctot, 0, cx, cx + cs, head[0])
            b_eff, b_cls = C.flow_fold_bias(_phys(wgt_), cout_, ctot, cx + cs, head[1], bias_)
            d2 = C.make_desc(list(srcs2), B, hh, ww, cout_, 3, 1, 1, True, act, module.mfma)
            w_f, _, w_f16, _ = _prep(d2, w_eff, False, want16=twins)
            return dict(w_eff=w_eff, w_fwd=w_f, w_fwd16=w_f16, bias=b_eff, bias_cls=b_cls,
                        cx=cx, cf_off=cx + cs, ctot=ctot, head=head)

        # Decoder stages whose third member is the previous stage's flow run, when
        # training, on cat[x, skip] with that member folded into weight space
        # (csrc/flowfold.hip; DVSOF_FLOW_FOLD=0: the member as a member)
        fold_pre = want_grad and os.environ.get('DVSOF_FLOW_FOLD', '1') != '0' and \
            len(module._extra_streams(dev)) == 0
        if side is not None or fold_pre:
            h16, w16 = H // 16, W // 16
            specs = [(4 + 2 * i + j, [(x0, 512, C.NHWC)], h16, w16, 512,
                      res[i][2 * j], False)
                     for i in range(NUM_RES) for j in range(2)]
            cx_, hh, ww = 512, h16, w16
            for i in range(4):
                srcs_ = [(x0, cx_, C.NHWC), (x0, ENC_CH[3 - i], C.NHWC)]
                if i > 0:
                    srcs_.append((x0, 2, C.NCHW))
                specs.append((4 + 2 * NUM_RES + i, srcs_, hh, ww, DEC_CH[i],
                              dec[i][0], True))
                cx_, hh, ww = DEC_CH[i], 2 * hh, 2 * ww
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                for li, srcs_, hh, ww, cout_, wgt_, up_ in specs:
                    if fold_pre and len(srcs_) == 3:
                        # decoder stage with a flow member: forward AND backward run
                        # on cat[x, skip] with the head folded into the weights
                        i_ = li - (4 + 2 * NUM_RES)
                        pre[li] = _fold_pre(srcs_[:2], hh, ww, cout_, wgt_, dec[i_][1],
                                            (dec[i_ - 1][2], dec[i_ - 1][3]))
                        continue
                    if side is None:
                        continue        # the other forms are made where they are used
                    d_ = C.make_desc(srcs_, B, hh, ww, cout_, 3, 1, 1, up_, act,
                                     module.mfma)
                    w_f, _, w_f16, _ = _prep(d_, _phys(wgt_), False,
                                                 want16=twins)
                    if w_f is not wgt_:
                        pre[li] = (w_f, w_f16)
            if side is not None:
                pre_ready = torch.cuda.Event()
                pre_ready.record(side)
        waited, waited16 = [False], [False]

        chain = {'v': None}     # Winograd form a layer made for its consumer
        # (Measured and not kept: the data-gradient weight forms issued late, beside the
        # decoder's kernels instead of layer by layer beside the encoder / residual layers:
        # 2.522-2.530 ms wherever they run -- the step is bound by total work, and the forms
        # cost 63 us of it: 2.451 ms with stale forms, DVSOF_STALE_FORMS=1.)

        def run(srcs, h, w, cout, wgt, bias, stride=1, up=False,
                residual=None, chained=(False, False)):
            """-> (y, y16): the layer's output and, in the bf16-twins mode,
            its bf16 copy (written by the same kernel).  chained = (its
            producer / its consumer is a Winograd layer of the same frame:
            the transformed input travels from output transform to GEMM,
            dvsof_conv_desc_t.winograd_next)."""
            d = C.make_desc(srcs, B, h, w, cout, 3, stride, 1, up, act,
                            module.mfma)
            # prepared weights: sub-pixel phase kernels for the decoder,
            # Winograd forms for the residual layers, and (when training) the
            # data-gradient form, made once per step
            first = len(L) == 0       # voxel input needs no data gradient
            need_dg = want_grad and not first
            w_fwd16 = w_dg16 = w_dg = fold = None
            fpre = pre.get(len(L)) if isinstance(pre.get(len(L)), dict) else None
            if fpre is not None:
                # flow member folded (forward and backward): two vector members
                if side is not None and not waited[0]:
                    main.wait_event(pre_ready)
                    waited[0] = True
                d2 = C.make_desc(list(srcs[:2]), B, h, w, cout, 3, 1, 1, True, act,
                                 module.mfma)
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    _, wd, _, wd16 = _prep(d2, fpre['w_eff'], True,
                                           phase_weights=fpre['w_fwd'], want16=twins)
                fold = dict(desc=d2, w_dg=wd, w_dg16=wd16, keep=(fpre,), cx=fpre['cx'],
                            cf_off=fpre['cf_off'], ctot=fpre['ctot'], head=fpre['head'])
                y, z = C.conv_fwd(d2, fpre['w_fwd'], fpre['bias'], dev, residual, mish,
                                  keep_input_transform=want_grad,
                                  weight16=fpre['w_fwd16'], bias_cls=fpre['bias_cls'])
                L.append(dict(desc=d, y=y, z=z, srcs=srcs, w=wgt, w_dg=None,
                              w_dg16=None, fold=fold))
                return y, d2._y16
            if side is not None and need_dg:
                if isinstance(pre.get(len(L)), tuple):    # made on the second stream
                    w_fwd, w_fwd16 = pre[len(L)]
                    if not waited[0]:
                        main.wait_event(pre_ready)
                        waited[0] = True
                else:
                    w_fwd, _ = C.prepare(d, _phys(wgt), False)
                    if w_fwd is not wgt:      # prepared form made on main
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                with torch.cuda.stream(side):
                    _, w_dg, _, w_dg16 = _prep(d, _phys(wgt), True,
                                               phase_weights=w_fwd,
                                               want16=twins)
            else:
                w_fwd, w_dg, w_fwd16, w_dg16 = _prep(
                    d, _phys(wgt), need_dg, want16=twins)
            if twins and w_fwd16 is None:
                if raw16_ready is not None and not waited16[0]:
                    main.wait_event(raw16_ready)
                    waited16[0] = True
                w_fwd16 = raw16.get(id(wgt))
                if w_fwd16 is None or w_fwd is not wgt:
                    w_fwd16 = C.to_bf16(w_fwd)
            v_pre = chain['v'] if chained[0] else None
            chain['v'] = C.winograd_form(d, cout, dev) \
                if chained[1] and C.winograd_chain(d, 0) else None
            y, z = C.conv_fwd(d, w_fwd, bias, dev, residual, mish,
                              keep_input_transform=want_grad,
                              weight16=w_fwd16, wino_pre=v_pre,
                              wino_next=chain['v'])
            L.append(dict(desc=d, y=y, z=z, srcs=srcs, w=wgt, w_dg=w_dg,
                          w_dg16=w_dg16, fold=fold))
            return y, d._y16

        # encoder (activations travel as (f32 tensor, bf16 twin or None))
        e, h, w = [], H, W
        cur, ccur, lay = (x0, None), Cin, C.NCHW
        for i in range(4):
            cur = run([(cur[0], ccur, lay, cur[1])], h, w, ENC_CH[i], *enc[i],
                      stride=2)
            h, w, ccur, lay = h // 2, w // 2, ENC_CH[i], C.NHWC
            e.append(cur)
        # residual blocks
        r = e[3]
        for i in range(NUM_RES):
            t = run([(t_[0], 512, C.NHWC, t_[1]) for t_ in (r,)], h, w, 512,
                    res[i][0], res[i][1], chained=(i > 0, True))
            r = run([(t[0], 512, C.NHWC, t[1])], h, w, 512, res[i][2],
                    res[i][3], residual=r[0], chained=(True, i + 1 < NUM_RES))
        # decoder
        flows, xx, cx, f, heads = [], r, 512, None, []
        for i in range(4):
            sk = e[3 - i]
            srcs = [(xx[0], cx, C.NHWC, xx[1]),
                    (sk[0], ENC_CH[3 - i], C.NHWC, sk[1])]
            if f is not None:
                srcs.append((f, 2, C.NCHW))
            xx = run(srcs, h, w, DEC_CH[i], dec[i][0], dec[i][1], up=True)
            x = xx[0]
            h, w, cx = 2 * h, 2 * w, DEC_CH[i]
            if fold_pre:
                # folded: no later stage reads this flow -- all four heads run in
                # one launch after the last stage (the tensor is only named here)
                f = torch.empty(B, 2, h, w, dtype=torch.float32, device=dev)
                heads.append((x, dec[i][2], dec[i][3], h, w, cx))
            else:
                f = C.head_fwd(x, dec[i][2], dec[i][3], B, h, w, cx)
            flows.append(f)
        if heads:
            C.heads_fwd(heads, B, out=flows)
        if want_grad:
            ctx.dg_ready = None
            if side is not None:
                ctx.dg_ready = torch.cuda.Event()
                ctx.dg_ready.record(side)
            ctx.L, ctx.act, ctx.dims = L, act, (B, Cin, H, W)
            ctx.raw16 = raw16       # made on the second stream, read on this one: alive until backward
            ctx.params = params
            ctx.module = module
        return tuple(flows)

    @staticmethod
    def backward(ctx, *gflows):
        L, act, (B, Cin, H, W) = ctx.L, ctx.act, ctx.dims
        params = ctx.params
        dev = gflows[0].device
        mish = act == C.ACT_MISH
        grads, finish = ctx.module._grad_targets(params)
        gflows = [g.contiguous() for g in gflows]

        def asrc(layer):        # what act' is evaluated on
            return layer['z'] if mish else layer['y']

        def wt(layer):
            return layer['w_dg']

        def new(t):
            return torch.empty_like(t)

        twins = ctx.module.mfma == C.MFMA_BF16_TWINS

        def tw(t):          # bf16 twin of a gradient a later data gradient reads
            return C.twin(t) if twins else None

        enc_l, res_l, dec_l = L[0:4], L[4:4 + 2 * NUM_RES], L[4 + 2 * NUM_RES:]
        po_res, po_dec = 8, 8 + 4 * NUM_RES

        # The data-gradient chain is the critical path; every weight gradient
        # only needs its layer's output gradient.  They run on a second HIP
        # stream so that their split-K reduces, bias sums and small-grid tails
        # fill the CUs the dgrad kernels leave idle.  ``keep`` holds the
        # tensors the side stream reads until the streams are joined.
        main = torch.cuda.current_stream(dev)
        side = ctx.module._wgrad_stream(dev)
        if ctx.dg_ready is not None:
            main.wait_event(ctx.dg_ready)
        keep = []

        # DVSOF_WGRAD_STREAMS=n (default 1): weight gradients dealt round-robin
        # to n side streams (they are independent of each other)
        sides = [side] + (ctx.module._extra_streams(dev) if side is not None else [])
        turn = [0]

        # Flow member folded into weight space (csrc/flowfold.hip): a stage's
        # weight gradient leaves its flow columns to dvsof_flow_fold_grads, which
        # also ADDS the stage's share to the gradients of the head below -- so it
        # runs once that head's own backward has written them: at the next
        # weight gradient, ahead of it on the same stream.
        pending = []

        def wgrad(desc, gz, gw, gb, unit, gz16=None, fold=None, first=None,
                  wino_gout=None):
            def body():
                if first is not None:
                    first()
                for job in pending:
                    job()
                del pending[:]
                C.conv_wgrad(desc, gz, gw, gb, gz16, skip_flat=fold is not None,
                             wino_gout=wino_gout)
                if fold is None:
                    finish(unit)
                else:
                    pending.append(fold_job(desc, gz, gw, gb, unit, fold))
            if side is None:
                body()
                return
            s_ = sides[turn[0] % len(sides)]
            turn[0] += 1
            ready = torch.cuda.Event()
            ready.record(main)
            s_.wait_event(ready)
            with torch.cuda.stream(s_):
                body()
            keep.extend((gz, gz16))

        def fold_job(desc, gz, gw, gb, unit, fold):
            ho, wo = C.out_size(desc)
            wh, bh = fold['head']
            i_h = fold['head_idx']

            def job():
                C.flow_fold_grads(gw, _phys(fold['w']), desc.Cout, fold['ctot'], 0,
                                  fold['cx'], fold['cf_off'], wh, bh, gb, gz, B, ho,
                                  wo, grads[i_h], grads[i_h + 1])
                finish(unit)
            return job

        # ---- decoder, fine to coarse
        g_x = None          # gradient w.r.t. dec[i].y from the finer stage
        g_f = gflows[3]     # total gradient of the flow of this stage
        g_skip = [None] * 4  # gradient into e[k] from the decoder
        g_r = g_r16 = None
        head_in_dgrad = False
        head_part = g_x16 = g_in16 = None
        fuse_heads = os.environ.get('DVSOF_NO_HEAD_FUSE', '0') == '0'
        fuse_general = os.environ.get('DVSOF_HEAD_FUSE_GENERAL', '0') == '1'
        # (the head's weight gradient from per-block partials of the same epilogue instead of
        # its own pass over the tensor on the other stream, dvsof_grad_dst_t.head_part: measured
        # 2.58 against 2.53 ms wherever the encoder's weight gradients go -- the shuffles, the
        # extra barrier and the partial stores sit on the data-gradient chain; DVSOF_HEAD_PARTS=1)
        head_parts = os.environ.get('DVSOF_HEAD_PARTS', '0') == '1'
        for i in (3, 2, 1, 0):
            lay = dec_l[i]
            d = lay['desc']
            h, w = C.out_size(d)
            y = lay['y']
            pw, pb, pfw, pfb = (po_dec + 4 * i + j for j in range(4))
            head_w = None
            if head_in_dgrad:
                # the finer stage's data gradient already wrote d/d(pre-activation)
                # of this stage (head path and act' in its epilogue): only the
                # head's own weight / bias gradient is left, off the critical chain
                gz, gz16 = g_x, g_x16

                if head_part is not None:
                    # ... of which that epilogue left per-block partial sums
                    def head_w(part=head_part, gw=grads[pfw], gb=grads[pfb], c=d.Cout):
                        C.head_reduce(part, c, gw, gb)
                else:
                    def head_w(y=y, wf=params[pfw], g=g_f, gw=grads[pfw], gb=grads[pfb],
                               h=h, w=w, c=d.Cout):
                        C.head_bwd(y, wf, g, None, None, act, None, gw, gb, B, h, w, c)
            else:
                gz, gz16 = new(y), tw(y)
                C.head_bwd(y, params[pfw], g_f, g_x, asrc(lay), act, gz,
                           grads[pfw], grads[pfb], B, h, w, d.Cout, gx16=gz16)
            fold = lay.get('fold')
            if fold is not None:
                fold = dict(fold, w=params[pw], head_idx=po_dec + 4 * (i - 1) + 2)
            wgrad(d, gz, grads[pw], grads[pb], ('dec', i), gz16, fold, first=head_w)
            srcs = lay['srcs']
            g_in = new(srcs[0][0])
            g_e = new(srcs[1][0])
            dsts = [dict(p=g_in), dict(p=g_e)]
            head_in_dgrad = False
            if i == 0:
                # x = r (last residual output): single consumer -> its dz,
                # which the residual chain's data gradients read next
                dsts[0]['actsrc'] = asrc(res_l[-1])
                g_r16 = dsts[0]['p16'] = tw(g_in)
            if fold is not None:
                # cat[x, skip] with the folded weights: g_in already holds the
                # path through the flow head, which then sees the loss gradient only
                g_fprev = gflows[i - 1]
                wf_prev = params[po_dec + 4 * (i - 1) + 2]
                # (the general kernels fold a head too, conv_epilogue: bf16s 5 146 against 5 234
                # samples/s without, bf16 3 941 / 3 912, bf16x3 the same -- their epilogue cannot
                # start its loads ahead of the K loop's end; DVSOF_HEAD_FUSE_GENERAL=1)
                if (fuse_heads and C.dgrad_fuses_head(fold['desc'])
                        and (fuse_general or C.dgrad_head_rows(fold['desc']) > 0)
                        and wf_prev.data_ptr() % 16 == 0):
                    # ... and the head below goes into this data gradient's epilogue
                    # (dvsof_grad_dst_t.head_w): g_in leaves as dec[i-1]'s dz
                    head_in_dgrad = True
                    g_in16 = tw(g_in)       # (twins mode: the data gradient below reads its bf16 copy)
                    dsts[0].update(head_w=wf_prev, head_gflow=g_fprev,
                                   actsrc=asrc(dec_l[i - 1]), p16=g_in16)
                    head_part = C.dgrad_head_part(fold['desc'], dec_l[i - 1]['desc'].Cout, dev) \
                        if head_parts else None
                    if head_part is not None:
                        dsts[0].update(head_x=dec_l[i - 1]['y'], head_part=head_part)
                        keep.append(head_part)
                C.conv_dgrad(fold['desc'], fold['w_dg'], gz, dsts, act,
                             weight16=fold['w_dg16'], gout16=gz16)
            else:
                if len(srcs) == 3:
                    g_fprev = new(srcs[2][0])
                    dsts.append(dict(p=g_fprev, addend=gflows[i - 1]))
                C.conv_dgrad(d, wt(lay), gz, dsts, act, weight16=lay['w_dg16'],
                             gout16=gz16)
            keep.append(gz16)
            g_skip[3 - i] = g_e
            if i > 0:
                g_x, g_f = g_in, g_fprev
                g_x16 = g_in16 if head_in_dgrad else None
            else:
                g_r = g_in
        # ---- residual blocks (g_r is already d/d pre-activation)
        gs, gs16 = g_r, g_r16
        # Winograd forms along the chain (dvsof_conv_desc_t.winograd_next /
        # winograd_next_gout): a data gradient's output transform also makes the
        # transformed input of the data gradient below and the gradient form of
        # the weight gradient below
        v_pre = z_pre = None

        def forms(producer, consumer):
            if not C.winograd_chain(producer, 1):
                return None, None
            c_out = consumer.Cout
            v = C.winograd_form(producer, c_out, dev)
            z = C.winograd_form(producer, c_out, dev) \
                if C.winograd_tile(consumer, 2) == 4 else None
            keep.extend((v, z))
            return v, z
        for i in reversed(range(NUM_RES)):
            l1, l2 = res_l[2 * i], res_l[2 * i + 1]
            pw1, pb1, pw2, pb2 = (po_res + 4 * i + j for j in range(4))
            wgrad(l2['desc'], gs, grads[pw2], grads[pb2], ('res', i, 2), gs16,
                  wino_gout=z_pre)
            g_t, g_t16 = new(l1['y']), tw(l1['y'])
            v_n, z_n = forms(l2['desc'], l1['desc'])
            C.conv_dgrad(l2['desc'], wt(l2), gs,
                         [dict(p=g_t, actsrc=asrc(l1), p16=g_t16)], act,
                         weight16=l2['w_dg16'], gout16=gs16, wino_pre=v_pre,
                         wino_next=v_n, wino_next_gout=z_n)
            v_pre, z_pre = v_n, z_n
            wgrad(l1['desc'], g_t, grads[pw1], grads[pb1], ('res', i, 1), g_t16,
                  wino_gout=z_pre)
            below = res_l[2 * i - 1] if i > 0 else enc_l[3]
            g_prev, g_prev16 = new(below['y']), tw(below['y'])
            dst = dict(p=g_prev, addend=gs, actsrc=asrc(below), p16=g_prev16)
            if i == 0:
                dst['addend2'] = g_skip[3]      # dec.0's skip into e4
            v_n, z_n = forms(l1['desc'], below['desc']) if i > 0 else (None, None)
            C.conv_dgrad(l1['desc'], wt(l1), g_t, [dst], act,
                         weight16=l1['w_dg16'], gout16=g_t16, wino_pre=v_pre,
                         wino_next=v_n, wino_next_gout=z_n)
            v_pre, z_pre = v_n, z_n
            keep.extend((gs16, g_t16))
            gs, gs16 = g_prev, g_prev16
        # ---- encoder.  Its weight gradients are issued on the MAIN stream after
        # the last data gradient: by then the second stream still holds the
        # residual blocks' weight gradients, and the main stream would only
        # wait for it -- this way both streams drain the tail together.
        gz, gz16 = gs, gs16
        deferred = []
        # (measured, batch 8: exact f32 on one GPU with the nine-product decoder kernels 3; the
        # bf16 modes and every data-parallel rank 2)
        red_ = getattr(ctx.module, 'reducer', None)
        alone_ = red_ is None or not red_.active()   # (under the exchange marks 2 again: 2.72 vs 2.84 ms)
        n_side = int(os.environ.get('DVSOF_ENC_SIDE_FROM', '3' if ctx.module.mfma == 0 and alone_ else '2'))
        for i in (3, 2, 1, 0):
            lay = enc_l[i]
            item = (lay['desc'], gz, grads[2 * i], grads[2 * i + 1], ('enc', i),
                    gz16)
            if i >= n_side:
                wgrad(*item)            # second stream (it is about to run dry)
            else:
                deferred.append(item)   # main stream, after the last data gradient
            if i == 0:
                break
            below = enc_l[i - 1]
            g_prev, g_prev16 = new(below['y']), tw(below['y'])
            C.conv_dgrad(lay['desc'], wt(lay), gz,
                         [dict(p=g_prev, addend=g_skip[i - 1],
                               actsrc=asrc(below), p16=g_prev16)], act,
                         weight16=lay['w_dg16'], gout16=gz16)
            keep.append(gz16)
            gz, gz16 = g_prev, g_prev16
        for desc, g, gw, gb, unit, g16 in deferred:
            if side is None:
                wgrad(desc, g, gw, gb, unit, g16)
            else:
                C.conv_wgrad(desc, g, gw, gb, g16)
                finish(unit)
        if side is not None:
            for s_ in sides:
                main.wait_stream(s_)
        del keep
        ctx.L = None
        return (None,) * (3 + len(params))


class Predictor(nn.Module):
    def __init__(self, in_channels, activation=None, compute_dtype='f32'):
        super().__init__()
        self.in_channels = in_channels
        self.act = activation_id(activation)
        # 'f32': exact f32 matrix cores.  'bf16': conv operands rounded to bf16
        # in registers (v_mfma_f32_32x32x16_bf16), f32 accumulation; weights,
        # activations, gradients and optimizer state all stay f32 in memory.
        # 'bf16x3': operands split into bf16 hi + lo, three products (error
        # ~2^-16 per product instead of 2^-24): f32-like accuracy at the bf16
        # matrix rate.
        # 'bf16s': bf16 operands as in 'bf16', but the forward / data-gradient
        # kernels stream bf16 TWINS of the activations, gradients and prepared
        # weights through LDS (half the LDS-DMA bytes 'bf16' is bound by);
        # every f32 tensor is still written (weight gradients, heads and loss
        # read those), master weights / optimizer state stay f32.
        modes = {'f32': C.MFMA_F32, 'bf16': C.MFMA_BF16, 'bf16x3': C.MFMA_BF16X3,
                 'bf16s': C.MFMA_BF16_TWINS}
        assert compute_dtype in modes, compute_dtype
        self.compute_dtype = compute_dtype
        self.mfma = modes[compute_dtype]
        self.reducer = None      # parallel.GradReducer for data parallelism
        chans = (in_channels,) + ENC_CH
        self.enc = nn.ModuleList(
            _Named(conv=ConvParams(chans[i], chans[i + 1], 3))
            for i in range(4))
        self.res = nn.ModuleList(
            _Named(conv1=ConvParams(512, 512, 3), conv2=ConvParams(512, 512, 3))
            for _ in range(NUM_RES))
        dec_in = (512 + 512, 256 + 256 + 2, 128 + 128 + 2, 64 + 64 + 2)
        self.dec = nn.ModuleList(
            _Named(conv=ConvParams(dec_in[i], DEC_CH[i], 3),
                   flow=ConvParams(DEC_CH[i], 2, 1))
            for i in range(4))

    # Gradient buckets in the order the backward completes them (fine decoder
    # stages first, encoder last): persistent flat buffers that the wgrad
    # kernels write into, that ``p.grad`` views, that the DP reducer
    # all-reduces and that the fused optimizer reads.
    UNIT_PARAMS = {
        **{('dec', i): tuple(8 + 4 * NUM_RES + 4 * i + j for j in range(4))
           for i in range(4)},
        **{('res', i, 2): (8 + 4 * i + 2, 8 + 4 * i + 3) for i in range(NUM_RES)},
        **{('res', i, 1): (8 + 4 * i, 8 + 4 * i + 1) for i in range(NUM_RES)},
        **{('enc', i): (2 * i, 2 * i + 1) for i in range(4)},
    }
    BUCKETS = (
        (('dec', 3), ('dec', 2), ('dec', 1)),
        (('dec', 0),),
        (('res', 1, 2),), (('res', 1, 1),), (('res', 0, 2),), (('res', 0, 1),),
        (('enc', 3),),
        (('enc', 2), ('enc', 1), ('enc', 0)),
    )

    def _wgrad_stream(self, dev):
        """Second stream of the backward (None: DVSOF_WGRAD_STREAM=0)."""
        if os.environ.get('DVSOF_WGRAD_STREAM', '1') == '0':
            return None
        # ONE second stream per device, shared by every model of the process:
        # ROCclr deals streams round-robin onto the hardware queues, and a third
        # or fourth stream lands on the main stream's queue (the two backward
        # chains then alternate instead of overlapping: 3.4 vs 2.4 ms per step
        # measured on the third model built in one process)
        key = torch.device(dev).index if torch.device(dev).index is not None \
            else torch.cuda.current_device()
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
        return _SIDE_STREAMS[key]

    def mark_step_begin(self, dev):
        """Called by the model wrapper before it voxelises: the second stream's
        work of the coming forward (weight twins, prepared forms) depends on the
        optimizer's update only, and may start HERE instead of behind the
        voxeliser (78 us of small kernels off the forward's lane at batch 8)."""
        if os.environ.get('DVSOF_NO_STEP_BEGIN') or self._wgrad_stream(dev) is None:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._step_begin = ev

    def _take_step_begin(self):
        ev, self._step_begin = getattr(self, '_step_begin', None), None
        return ev

    def _extra_streams(self, dev):
        n = int(os.environ.get('DVSOF_WGRAD_STREAMS', '1')) - 1
        key = torch.device(dev).index if torch.device(dev).index is not None \
            else torch.cuda.current_device()
        have = _EXTRA_STREAMS.setdefault(key, [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=dev))
        return have[:max(n, 0)]

    def _buckets(self, params):
        dev = params[0].device
        key = (dev, tuple(p.numel() for p in params))
        if getattr(self, '_bucket_key', None) != key:
            self._bucket_key = key
            self._bucket_flat, self._bucket_slot = [], {}
            for b, units in enumerate(self.BUCKETS):
                off = 0
                for u in units:
                    for i in self.UNIT_PARAMS[u]:
                        self._bucket_slot[i] = (b, off)
                        off += params[i].numel()
                self._bucket_flat.append(
                    torch.empty(off, dtype=torch.float32, device=dev))
            self._bucket_views = None
        return self._bucket_flat, self._bucket_slot

    def attach_bucket_grads(self):
        """``p.grad`` = the parameter's slice of its gradient bucket wherever
        ``p.grad`` is unset.  After a captured micro-batch that only wrote or
        accumulated gradients (capture.py roles 'first' / 'middle') this is
        Python's view of what the replay did: the next eager micro-batch of
        the same optimizer step then accumulates into the buckets."""
        params = self.param_list()
        flats, slot = self._buckets(params)
        if getattr(self, '_bucket_views', None) is None:
            self._bucket_views = [
                flats[slot[i][0]][slot[i][1]:slot[i][1] + p.numel()]
                .as_strided(p.shape, p.stride()) for i, p in enumerate(params)]
        for p, v in zip(params, self._bucket_views):
            if p.grad is None:
                p.grad = v

    def _grad_targets(self, params):
        """-> (targets, finish): per-parameter tensors the backward writes
        into, and a callback to call when a unit's gradients are enqueued.
        A parameter without .grad gets a view of its bucket as .grad; one that
        already has a gradient (micro-batch accumulation,
        reference utils/training.py:156-167) is accumulated into."""
        flats, slot = self._buckets(params)
        views, targets, accumulate = [], [], []
        for i, p in enumerate(params):
            b, off = slot[i]
            v = flats[b][off:off + p.numel()].as_strided(p.shape, p.stride())
            views.append(v)
            if p.grad is None:
                targets.append(v)
                accumulate.append(False)
            else:
                targets.append(torch.empty_like(p))
                accumulate.append(True)
        remaining = [len(units) for units in self.BUCKETS]
        unit_bucket = {u: b for b, units in enumerate(self.BUCKETS)
                       for u in units}
        reducer = self.reducer
        dev, on_gpu = params[0].device, params[0].is_cuda
        # The units of one bucket may be enqueued on different streams (the
        # backward splits weight gradients between its two streams).  Whoever
        # closes the bucket hands it to the reducer / the fused optimizer on
        # ITS stream, so that stream first waits for the other units' writes.
        marks = [[] for _ in self.BUCKETS]

        def finish(unit):
            for i in self.UNIT_PARAMS[unit]:
                p = params[i]
                if accumulate[i]:
                    p.grad.add_(targets[i])
                else:
                    p.grad = views[i]
            b = unit_bucket[unit]
            remaining[b] -= 1
            if on_gpu and len(self.BUCKETS[b]) > 1:
                cur = torch.cuda.current_stream(dev)
                if remaining[b] != 0:
                    done = torch.cuda.Event()
                    done.record(cur)
                    marks[b].append((cur, done))
                else:
                    for stream, done in marks[b]:
                        if stream != cur:
                            cur.wait_event(done)
            if remaining[b] != 0:
                return
            hook = getattr(self, 'bucket_hook', None)    # optim.fuse_into_backward
            after = None
            if hook is not None:
                bparams = [params[i] for u in self.BUCKETS[b]
                           for i in self.UNIT_PARAMS[u]]
                after = lambda: hook(b, bparams)            # noqa: E731
            if reducer is not None and reducer.active():
                owned = all(params[i].grad.data_ptr() == views[i].data_ptr()
                            for u in self.BUCKETS[b]
                            for i in self.UNIT_PARAMS[u])
                assert owned, 'DP needs .grad to live in the bucket buffers'
                reducer.bucket_ready(flats[b], after)
            elif after is not None:
                after()
        return targets, finish

    def param_list(self):
        out = []
        for m in self.enc:
            out += [m.conv.weight, m.conv.bias]
        for m in self.res:
            out += [m.conv1.weight, m.conv1.bias, m.conv2.weight, m.conv2.bias]
        for m in self.dec:
            out += [m.conv.weight, m.conv.bias, m.flow.weight, m.flow.bias]
        return out

    def forward(self, voxels):
        """voxels: float32 [B,C,H,W] (NCHW dense), H and W multiples of 16.
        -> tuple of 4 flows [B,2,H/8..H,W/8..W] coarse to fine."""
        B, Cin, H, W = voxels.shape
        assert Cin == self.in_channels
        assert H % 16 == 0 and W % 16 == 0, \
            'the predictor needs H and W divisible by 16'
        params = self.param_list()
        want_grad = torch.is_grad_enabled() and any(p.requires_grad
                                                    for p in params)
        return _PredictorFn.apply(voxels.contiguous(), self, want_grad,
                                  *params)

    def flops_per_sample(self, H, W):
        """Algorithmic forward FLOPs (2*MACs), SURVEY.md section 8d formula."""
        total, h, w, c = 0, H, W, self.in_channels
        for co in ENC_CH:
            h, w = h // 2, w // 2
            total += 2 * h * w * co * c * 9
            c = co
        total += 2 * NUM_RES * 2 * h * w * 512 * 512 * 9
        cin = (1024, 514, 258, 130)
        for i, co in enumerate(DEC_CH):
            h, w = 2 * h, 2 * w
            total += 2 * h * w * co * cin[i] * 9 + 2 * h * w * 2 * co
        return total
