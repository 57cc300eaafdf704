"""The six Residual_blocks of the AASIST encoder (model/wav2vec2_aasist.py:377-433, stacked at :470-476) as ONE autograd node over the
kernels of csrc/resstack.hip: forward = 3 launches per block (conv1 + BatchNorm statistics, BatchNorm + SELU, conv2 + residual [+ the
1x3 down-sample convolution]), backward = 7-9 per block, no padded copies, no torch glue, parameter gradients written straight into
`p.grad` (the flat gradient buffer of the model) by the reducing kernels.

    res_stack(x0 [B, H, W, 1] f32 channels-last, blocks = [Residual_block, ...]) -> [B, H, W, C_last]

Maps live zero-bordered and flat (see the header of resstack.hip): a (kh, kw) tap is a flat shift, so the stride-1 convolutions, their
data gradients (negated shifts, transposed weights) and weight gradients all read the maps in place.

Reference quirks kept: conv1 reads the block INPUT (the bn1 + SELU result is discarded, :414-420) while bn1's running statistics are
still updated in training — here from the statistics the previous block's conv2 epilogue already gathers; conv1's bias gradient is
computed (it is round-off around zero: BatchNorm cancels the bias) because the reference computes it.
"""
import ctypes
import weakref

import torch

from . import lib as L
from . import ops

_SIZES = {(16, 32), (32, 32), (32, 64), (64, 64)}


MAX_W = 93      # RS_MAX_SPAN of resstack.hip: W + 3 positions of halo


def supported(blocks, W=None):
    """The instantiated channel steps: 1 (carried as 16) | 32 | 64 -> 32 | 64, kernels (2,3) / (2,3) / (1,3); maps up to 93 positions wide."""
    if W is not None and W > MAX_W:
        return False
    try:
        for i, blk in enumerate(blocks):
            co, ci, kh, kw = blk.conv1.weight.shape
            if (kh, kw) != (2, 3) or tuple(blk.conv2.weight.shape) != (co, co, 2, 3) or (max(ci, 16), co) not in _SIZES:
                return False
            if ci not in (1, 16, 32, 64) or (ci == 1 and i != 0):
                return False
            if blk.downsample and tuple(blk.conv_downsample.weight.shape) != (co, ci, 1, 3):
                return False
            if not blk.downsample and ci != co:
                return False
        return len(blocks) > 0
    except AttributeError:
        return False


def _grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class _Plan:
    """Buffers of one (B, H, W, channel plan); a plan is busy from a grad-mode forward until its backward."""

    def __init__(self, B, H, W, chans, dev):
        self.key = (B, H, W, tuple(chans))
        self.B, self.H, self.W, self.Wp = B, H, W, W + 2
        self.G = B * (H + 2) * self.Wp
        self.slack = self.Wp + 2 + 256 + 6          # positions on either side of [0, G): halo of the first / last (256-position) tile
        self.busy = False
        self.dev = dev
        z = lambda c: torch.zeros((self.G + 2 * self.slack) * c, device=dev)      # zero once: the slack is read, never written
        cps = [max(16, chans[0])] + list(chans[1:])
        self.cps = cps
        n = len(cps) - 1
        self.x = [z(c) for c in cps]
        self.y1 = [z(c) for c in cps[1:]]
        self.a = [z(c) for c in cps[1:]]
        self.tmp = z(max(cps))
        self.stats = [torch.zeros(4 * c, device=dev) for c in cps[1:]]
        self.bstats = torch.zeros(2 * max(cps), device=dev)
        self.acc = torch.zeros(16 * 2 * max(cps), dtype=torch.float64, device=dev)      # SCL_RS_NSLOT rows; left zeroed by every finishing block
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dx = [z(c) for c in cps]
        self.dz = z(max(cps))
        self.dtmp = z(max(cps))
        self.wpk = None
        self.sig, self.fwd_calls, self.bwd_calls, self.bwd_need = None, None, None, None
        nsl = L.load().scl_rs_wgrad_nslabs
        self.part = torch.empty(max(max(nsl(ci, co) * 6 * ci * co for ci, co in zip(cps[:-1], cps[1:])), max(nsl(c, c) * 6 * c * c for c in cps[1:])), device=dev)
        self.gx = L.SclRsGeom(B, H, W, 1, H, 0)        # block inputs / outputs: rows 1..H
        self.gy = L.SclRsGeom(B, H, W, 0, H, 0)        # conv1 outputs (H + 1 rows): rows 0..H
        Wp = self.Wp
        self.s1 = [kh * Wp + kw - 1 for kh in (0, 1) for kw in (0, 1, 2)]
        self.s2 = [(kh - 1) * Wp + kw - 1 for kh in (0, 1) for kw in (0, 1, 2)]
        self.sd = [kw - 1 for kw in (0, 1, 2)]

    def p(self, t, c):
        """Pointer to flat position 0 of a bordered map with c channels."""
        return t.data_ptr() + 4 * self.slack * c

    def attn_buffers(self):
        """Maps and statistics of the attention block behind the stack (all 64 channels wide, the 128-channel middle as two planes)."""
        if getattr(self, "at", None) is None:
            z = lambda: torch.zeros((self.G + 2 * self.slack) * 64, device=self.dev)
            f = lambda n: torch.zeros(n, device=self.dev)
            self.at = dict(xa=z(), a1=[z(), z()], b1=[z(), z()], wl=z(), dxa=z(), dwl=z(), db1=[z(), z()], st_fb=f(256), st_a=[f(256), f(256)],
                           bst_a=[f(128), f(128)], bst_fb=f(128), wpk=torch.empty(8 * 64 * 64, device=self.dev),
                           job=(L.SclGraphReduceJob * 1)())
        return self.at


_PLANS = []


def _acquire(B, H, W, chans, dev, hold):
    key = (B, H, W, tuple(chans))
    for pl in _PLANS:
        if pl.key == key and pl.dev == dev and not pl.busy:
            pl.busy = hold
            pl.gen = getattr(pl, "gen", 0) + 1
            return pl
    # never hand out a plan whose backward is still pending (its saved maps, statistics, dropout seeds and recorded pointers belong to
    # that node): a further forward of the same shape gets a fresh plan, up to a hard cap.  Plans of nodes that are freed without a
    # backward are released by the finalizer _hold() attaches to the node.
    same = [pl for pl in _PLANS if pl.key == key and pl.dev == dev]
    if len(same) >= MAX_LIVE_PLANS:
        raise RuntimeError("%d forward passes of shape %r are waiting for their backward; run the backwards (or drop the graphs) "
                           "before another forward of this shape" % (len(same), key))
    idle = [pl for pl in _PLANS if not pl.busy]
    if len(_PLANS) >= 8 and idle:
        _PLANS.remove(idle[0])
    pl = _Plan(B, H, W, chans, dev)
    pl.busy = hold
    pl.gen = 1
    _PLANS.append(pl)
    return pl


MAX_LIVE_PLANS = 16


def _storages(*ts):
    """What a plan keeps of the tensors its in-flight kernels read / write: the STORAGE (the memory stays allocated), never the tensor
    object — an output tensor carries its grad_fn, and a plan that held it would keep its own autograd node (and through ctx.pl itself)
    alive until the next forward of the shape, so a graph dropped without a backward could never hand its plan back."""
    out = []
    for t in ts:
        if isinstance(t, (tuple, list)):
            out.append(_storages(*t))
        elif torch.is_tensor(t):
            out.append(t.untyped_storage())
        else:
            out.append(t)
    return tuple(out)


def _release(pl, gen):
    if pl.gen == gen:
        pl.busy = False


def _hold(ctx, pl):
    """Ties the plan to the autograd node: generation stamp for the backward's ownership check, release when the node dies unused."""
    ctx.pl, ctx.gen = pl, pl.gen
    if pl.busy:
        weakref.finalize(ctx, _release, pl, pl.gen)


def _owned(ctx):
    pl = ctx.pl
    if pl.gen != ctx.gen:
        raise RuntimeError("this node's plan was re-used by a later forward (generation %d, node holds %d): its saved activations are gone" % (pl.gen, ctx.gen))
    return pl


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _conv(pl, inp, wpk, out, cin, cout, shifts, geom, *, bias=None, addend=None, stat_mode=0, nvalid=0.0, act_a=None, y1=None, bnstats=None,
          gamma=None, beta=None, run_mean=None, run_var=None, nbt=None, stats_out=None, dgamma=None, dbeta=None, training=True, eps=1e-5, momentum=0.1,
          epi_act=0):
    d = L.SclRsConv()
    d.inp, d.wpk, d.out = inp, wpk, out
    pt = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
    d.bias, d.addend, d.act_a, d.y1, d.bnstats = pt(bias), pt(addend), pt(act_a), pt(y1), pt(bnstats)
    d.acc, d.ticket = pl.acc.data_ptr(), pl.ticket.data_ptr()
    d.gamma, d.beta, d.run_mean, d.run_var, d.nbt = pt(gamma), pt(beta), pt(run_mean), pt(run_var), pt(nbt)
    d.stats_out, d.dgamma, d.dbeta = pt(stats_out), pt(dgamma), pt(dbeta)
    d.nvalid = float(nvalid)
    d.geom = geom
    for i, s in enumerate(shifts):
        d.shift[i] = s
    d.cin, d.cout, d.ntaps, d.stat_mode, d.training = cin, cout, len(shifts), stat_mode, 1 if training else 0
    d.eps, d.momentum, d.epi_act = eps, momentum, epi_act
    ops._call("scl_rs_conv", ctypes.byref(d), _stream(), keep=d)


def _pack_all(pl, blocks):
    """Register images of every convolution (forward and data gradient) in one launch, into plan-owned buffers."""
    jobs, views, off = [], {}, 0
    total = 0
    spec = []
    for i, blk in enumerate(blocks):
        ci_p, co = pl.cps[i], pl.cps[i + 1]
        spec.append((i, "c1", blk.conv1.weight, 6, ci_p, co, 0))
        spec.append((i, "c2", blk.conv2.weight, 6, co, co, 0))
        spec.append((i, "c1T", blk.conv1.weight, 6, co, ci_p, 1))
        spec.append((i, "c2T", blk.conv2.weight, 6, co, co, 1))
        if blk.downsample:
            spec.append((i, "ds", blk.conv_downsample.weight, 3, ci_p, co, 0))
            spec.append((i, "dsT", blk.conv_downsample.weight, 3, co, ci_p, 1))
    sizes = [nt * cinp * coutp for (_, _, _, nt, cinp, coutp, _) in spec]
    if pl.wpk is None or pl.wpk.numel() < sum(sizes):
        pl.wpk = torch.empty(sum(sizes), device=pl.dev)
    arr = (L.SclRsPackJob * len(spec))()
    for k, ((i, name, w, nt, cinp, coutp, tr), n) in enumerate(zip(spec, sizes)):
        wc = w.detach()
        assert wc.is_contiguous() and wc.dtype == torch.float32
        arr[k] = L.SclRsPackJob(wc.data_ptr(), pl.wpk.data_ptr() + 4 * off, wc.shape[0], wc.shape[1], nt, cinp, coutp, tr)
        views[(i, name)] = pl.wpk.data_ptr() + 4 * off
        off += n
    assert len(spec) <= 32, "one pack launch holds 32 images (six Residual_blocks need 28)"
    ops._call("scl_rs_pack_weights", arr, len(spec), _stream(), keep=arr)
    return views


def _bn_args(bn):
    return dict(eps=float(bn.eps), momentum=0.1 if bn.momentum is None else float(bn.momentum))


def _signature(blocks, training):
    """What a plan's recorded launch sequence depends on besides its own buffers: parameter / gradient / buffer storage and the mode."""
    w0, wl = blocks[0].conv1.weight, blocks[-1].conv2.weight
    return (torch.cuda.current_stream().cuda_stream, id(blocks[0]), len(blocks), training, w0.data_ptr(), wl.data_ptr(), _grad(w0).data_ptr(), _grad(wl).data_ptr(),
            _grad(blocks[-1].bn2.bias).data_ptr(), blocks[-1].bn2.running_mean.data_ptr())


def _forward(pl, x0, blocks, training, attn=None):
    """The first call of a (plan, parameter storage, mode) runs the launch sequence while ops records it; later calls patch the input /
    output pointers and replay the recorded C calls."""
    B, H, W = pl.B, pl.H, pl.W
    S = _stream
    c_last = pl.cps[-1]
    out = torch.empty(B, H, W, c_last, device=pl.dev) if attn is None else (torch.empty(B, H, c_last, device=pl.dev), torch.empty(B, W, c_last, device=pl.dev))
    sig = _signature(blocks, training) + ((id(attn[0]), attn[2].data_ptr(), _grad(attn[2]).data_ptr(), _grad(attn[1][0].weight).data_ptr()) if attn is not None else ())
    if pl.sig == sig and pl.fwd_calls is not None:
        pl.in_entry[1][0] = x0.data_ptr()
        if attn is None:
            pl.out_entry[1][1] = out.data_ptr()
        else:
            pl.pool_entry[1][3], pl.pool_entry[1][4] = out[0].data_ptr(), out[1].data_ptr()
        pl.live = _storages(x0, out)
        ops.replay(pl.fwd_calls)
        return out
    pl.sig, pl.fwd_calls, pl.bwd_calls = sig, None, None
    record = ops._rec() is None
    if record:
        ops.start_recording()
    wv = _pack_all(pl, blocks)
    pl.wv = wv
    pl.in_entry = ops._call("scl_rs_copy", x0.data_ptr(), pl.p(pl.x[0], pl.cps[0]), x0.shape[-1], pl.cps[0], 0, ctypes.byref(pl.gx), S())
    n_y, n_x = float(B * (H + 1) * W), float(B * H * W)
    for i, blk in enumerate(blocks):
        ci, co = pl.cps[i], pl.cps[i + 1]
        xin, y1, a, xout = pl.p(pl.x[i], ci), pl.p(pl.y1[i], co), pl.p(pl.a[i], co), pl.p(pl.x[i + 1], co)
        bn2 = blk.bn2
        if training:
            _conv(pl, xin, wv[(i, "c1")], y1, ci, co, pl.s1, pl.gy, bias=blk.conv1.bias, stat_mode=1, nvalid=n_y, gamma=bn2.weight, beta=bn2.bias,
                  run_mean=bn2.running_mean, run_var=bn2.running_var, nbt=bn2.num_batches_tracked, stats_out=pl.stats[i], **_bn_args(bn2))
        else:
            _conv(pl, xin, wv[(i, "c1")], y1, ci, co, pl.s1, pl.gy, bias=blk.conv1.bias)
            ops._call("scl_rs_bn_eval_stats", bn2.weight.data_ptr(), bn2.bias.data_ptr(), bn2.running_mean.data_ptr(), bn2.running_var.data_ptr(),
                      float(bn2.eps), co, pl.stats[i].data_ptr(), S())
        ops._call("scl_rs_bn_act", y1, pl.stats[i].data_ptr(), a, co, 1, ctypes.byref(pl.gy), S())
        if blk.downsample:
            tmp = pl.p(pl.tmp, co)
            _conv(pl, xin, wv[(i, "ds")], tmp, ci, co, pl.sd, pl.gx, bias=blk.conv_downsample.bias)
            addend = tmp
        else:
            addend = xin
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        if training and nxt is not None and not nxt.first:      # the next block's bn1: result discarded by the reference, running statistics kept
            bn1 = nxt.bn1
            _conv(pl, a, wv[(i, "c2")], xout, co, co, pl.s2, pl.gx, bias=blk.conv2.bias, addend=addend, stat_mode=1, nvalid=n_x,
                  run_mean=bn1.running_mean, run_var=bn1.running_var, nbt=bn1.num_batches_tracked, **_bn_args(bn1))
        elif training and nxt is None and attn is not None:      # first_bn1 (model/wav2vec2_aasist.py:520) reads the stack output: its statistics ride here
            fb1 = attn[0]
            _conv(pl, a, wv[(i, "c2")], xout, co, co, pl.s2, pl.gx, bias=blk.conv2.bias, addend=addend, stat_mode=1, nvalid=n_x, gamma=fb1.weight, beta=fb1.bias,
                  run_mean=fb1.running_mean, run_var=fb1.running_var, nbt=fb1.num_batches_tracked, stats_out=pl.attn_buffers()["st_fb"], **_bn_args(fb1))
        else:
            _conv(pl, a, wv[(i, "c2")], xout, co, co, pl.s2, pl.gx, bias=blk.conv2.bias, addend=addend)
    if attn is None:
        pl.out_entry = ops._call("scl_rs_copy", pl.p(pl.x[-1], c_last), out.data_ptr(), c_last, c_last, 1, ctypes.byref(pl.gx), S())
    else:
        _attn_forward(pl, attn, training, out)
    pl.live = _storages(x0, out)
    if record:
        pl.fwd_calls = ops.stop_recording()
    return out


def _attn_pack(pl, attn):
    """Register images of the two 1x1 convolutions of the attention block, as 64 -> 64 planes: W1 [128, 64] by output rows, W3 [64, 128]
    by input columns; forward and data-gradient images."""
    at = pl.attn_buffers()
    w1, w3 = attn[1][0].weight, attn[1][3].weight
    arr = (L.SclRsPackJob * 8)()
    views = {}
    k = 0
    for p in range(2):
        for name, w, off, ld, tr in (("w1", w1, 64 * 64 * p, 64, 0), ("w1T", w1, 64 * 64 * p, 64, 1), ("w3", w3, 64 * p, 128, 0), ("w3T", w3, 64 * p, 128, 1)):
            dst = at["wpk"].data_ptr() + 4 * k * 64 * 64
            arr[k] = L.SclRsPackJob(w.data_ptr() + 4 * off, dst, 64, 64, 1, 64, 64, tr, ld, 0)
            views[(name, p)] = dst
            k += 1
    ops._call("scl_rs_pack_weights", arr, 8, _stream(), keep=arr)
    return views


def _attn_forward(pl, attn, training, out):
    """first_bn1 + SELU, attention = conv(64 -> 128) -> SELU -> BatchNorm(128) -> conv(128 -> 64), the two soft-max poolings (+ pos_S)
    (model/wav2vec2_aasist.py:520-541) on the bordered maps; out = (e_S, e_T)."""
    fb1, (a0, bnA, a3), pos_S = attn[0], (attn[1][0], attn[1][2], attn[1][3]), attn[2]
    B, H, W = pl.B, pl.H, pl.W
    S = _stream
    at = pl.attn_buffers()
    wv = _attn_pack(pl, attn)
    pl.awv = wv
    n_x = float(B * H * W)
    P = lambda t: pl.p(t, 64)
    x6 = P(pl.x[-1])
    if not training:
        ops._call("scl_rs_bn_eval_stats", fb1.weight.data_ptr(), fb1.bias.data_ptr(), fb1.running_mean.data_ptr(), fb1.running_var.data_ptr(), float(fb1.eps), 64,
                  at["st_fb"].data_ptr(), S())
    ops._call("scl_rs_bn_act", x6, at["st_fb"].data_ptr(), P(at["xa"]), 64, 1, ctypes.byref(pl.gx), S())
    for p in range(2):
        o = 4 * 64 * p
        if training:
            _conv(pl, P(at["xa"]), wv[("w1", p)], P(at["a1"][p]), 64, 64, [0], pl.gx, bias=a0.bias.data_ptr() + o, stat_mode=1, nvalid=n_x,
                  gamma=bnA.weight.data_ptr() + o, beta=bnA.bias.data_ptr() + o, run_mean=bnA.running_mean.data_ptr() + o, run_var=bnA.running_var.data_ptr() + o,
                  nbt=bnA.num_batches_tracked if p == 0 else None, stats_out=at["st_a"][p], epi_act=1, **_bn_args(bnA))
        else:
            _conv(pl, P(at["xa"]), wv[("w1", p)], P(at["a1"][p]), 64, 64, [0], pl.gx, bias=a0.bias.data_ptr() + o, epi_act=1)
            ops._call("scl_rs_bn_eval_stats", bnA.weight.data_ptr() + o, bnA.bias.data_ptr() + o, bnA.running_mean.data_ptr() + o, bnA.running_var.data_ptr() + o,
                      float(bnA.eps), 64, at["st_a"][p].data_ptr(), S())
        ops._call("scl_rs_bn_act", P(at["a1"][p]), at["st_a"][p].data_ptr(), P(at["b1"][p]), 64, 0, ctypes.byref(pl.gx), S())
    _conv(pl, P(at["b1"][0]), wv[("w3", 0)], P(at["wl"]), 64, 64, [0], pl.gx, bias=a3.bias)
    _conv(pl, P(at["b1"][1]), wv[("w3", 1)], P(at["wl"]), 64, 64, [0], pl.gx, addend=P(at["wl"]))
    pl.pool_entry = ops._call("scl_rs_attn_pool_fwd", P(at["xa"]), P(at["wl"]), pos_S.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), 64, ctypes.byref(pl.gx), S())


def _attn_backward(pl, attn, training, deS, deT):
    """Gradients of the attention block and of first_bn1 + SELU; leaves d(stack output) in pl.dx[-1]."""
    fb1, (a0, bnA, a3), pos_S = attn[0], (attn[1][0], attn[1][2], attn[1][3]), attn[2]
    B, H, W = pl.B, pl.H, pl.W
    S = _stream
    at, wv = pl.at, pl.awv
    n_x = float(B * H * W)
    P = lambda t: pl.p(t, 64)
    x6, dz6 = P(pl.x[-1]), P(pl.dx[-1])
    pl.poolb_entry = ops._call("scl_rs_attn_pool_bwd", P(at["xa"]), P(at["wl"]), deS.data_ptr(), deT.data_ptr(), P(at["dxa"]), P(at["dwl"]), 64, ctypes.byref(pl.gx), S())
    job = at["job"]
    job[0] = L.SclGraphReduceJob(deS.data_ptr(), _grad(pos_S).data_ptr(), H * 64, H * 64, B)      # d pos_S = sum_b d e_S
    ops._call("scl_graph_reduce", job, 1, S(), keep=job)
    for p in range(2):
        o = 4 * 64 * p
        _conv(pl, P(at["dwl"]), wv[("w3T", p)], P(at["db1"][p]), 64, 64, [0], pl.gx, stat_mode=2, nvalid=n_x, y1=P(at["a1"][p]), bnstats=at["st_a"][p],
              dgamma=_grad(bnA.weight).data_ptr() + o, dbeta=_grad(bnA.bias).data_ptr() + o, stats_out=at["bst_a"][p], training=training)
        _wgrad(pl, P(at["b1"][p]), P(at["dwl"]), 64, 64, [0], a3.weight, a3.bias if p == 0 else None, 64, 64, ld=128, w_off=64 * p)
        ops._call("scl_rs_bn_bwd_apply", P(at["db1"][p]), P(at["a1"][p]), at["st_a"][p].data_ptr(), at["bst_a"][p].data_ptr(), 64, 1, ctypes.byref(pl.gx), S())
        _wgrad(pl, P(at["xa"]), P(at["db1"][p]), 64, 64, [0], a0.weight, a0.bias, 64, 64, ld=64, w_off=64 * 64 * p, bias_off=64 * p)
    _conv(pl, P(at["db1"][0]), wv[("w1T", 0)], dz6, 64, 64, [0], pl.gx, addend=P(at["dxa"]))
    _conv(pl, P(at["db1"][1]), wv[("w1T", 1)], dz6, 64, 64, [0], pl.gx, addend=dz6, stat_mode=2, nvalid=n_x, act_a=P(at["xa"]), y1=x6, bnstats=at["st_fb"],
          dgamma=_grad(fb1.weight), dbeta=_grad(fb1.bias), stats_out=at["bst_fb"], training=training)
    ops._call("scl_rs_bn_bwd_apply", dz6, x6, at["st_fb"].data_ptr(), at["bst_fb"].data_ptr(), 64, 0, ctypes.byref(pl.gx), S())


def _wgrad(pl, inp, dout, cin, cout, shifts, w, bias, co_real, ci_real, ld=0, w_off=0, bias_off=0):
    arr = (ctypes.c_int32 * 6)(*(list(shifts) + [0] * (6 - len(shifts))))
    nslab = L.load().scl_rs_wgrad_nslabs(cin, cout)
    ops._call("scl_rs_wgrad", inp, dout, cin, cout, len(shifts), arr, ctypes.byref(pl.gx), pl.part.data_ptr(), pl.acc.data_ptr(), pl.ticket.data_ptr(),
              None if bias is None else _grad(bias).data_ptr() + 4 * bias_off, _stream(), keep=arr)
    ops._call("scl_rs_wgrad_reduce", pl.part.data_ptr(), nslab, len(shifts), cin, cout, co_real, ci_real, ld, _grad(w).data_ptr() + 4 * w_off, _stream())


def _backward(pl, d_out, blocks, training, need_dx0, attn=None):
    B, H, W = pl.B, pl.H, pl.W
    S = _stream
    wv = pl.wv
    c_last = pl.cps[-1]
    n_y = float(B * (H + 1) * W)
    d_out = d_out.contiguous().float() if attn is None else (d_out[0].contiguous().float(), d_out[1].contiguous().float())
    dx0 = torch.empty(B, H, W, 1, device=pl.dev) if need_dx0 else None
    if pl.bwd_calls is not None and pl.bwd_need == need_dx0 and pl.sig[: len(_signature(blocks, training))] == _signature(blocks, training):
        if attn is None:
            pl.din_entry[1][0] = d_out.data_ptr()
        else:
            pl.poolb_entry[1][2], pl.poolb_entry[1][3] = d_out[0].data_ptr(), d_out[1].data_ptr()
            pl.at["job"][0].src = d_out[0].data_ptr()
        if need_dx0:
            pl.dout_entry[1][1] = dx0.data_ptr()
        pl.live_b = _storages(d_out, dx0)
        ops.replay(pl.bwd_calls)
        return dx0
    record = ops._rec() is None and pl.fwd_calls is not None
    if record:
        ops.start_recording()
    if attn is None:
        pl.din_entry = ops._call("scl_rs_copy", d_out.data_ptr(), pl.p(pl.dx[-1], c_last), c_last, c_last, 0, ctypes.byref(pl.gx), S())
    else:
        _attn_backward(pl, attn, training, d_out[0], d_out[1])
    neg = lambda s: [-v for v in s]
    for i in reversed(range(len(blocks))):
        blk = blocks[i]
        ci, co = pl.cps[i], pl.cps[i + 1]
        ci_real = blk.conv1.weight.shape[1]
        dcur, dprev = pl.p(pl.dx[i + 1], co), pl.p(pl.dx[i], ci)
        xin, y1, a, dz = pl.p(pl.x[i], ci), pl.p(pl.y1[i], co), pl.p(pl.a[i], co), pl.p(pl.dz, co)
        bn2 = blk.bn2
        # conv2 data gradient x selu'(a) -> dz, its two batch sums -> dgamma / dbeta / the means of the BatchNorm backward
        _conv(pl, dcur, wv[(i, "c2T")], dz, co, co, neg(pl.s2), pl.gy, stat_mode=2, nvalid=n_y, act_a=a, y1=y1, bnstats=pl.stats[i],
              dgamma=_grad(bn2.weight), dbeta=_grad(bn2.bias), stats_out=pl.bstats, training=training)
        ops._call("scl_rs_bn_bwd_apply", dz, y1, pl.stats[i].data_ptr(), pl.bstats.data_ptr(), co, 0, ctypes.byref(pl.gy), S())      # dz is now d(conv1 output)
        _wgrad(pl, a, dcur, co, co, pl.s2, blk.conv2.weight, blk.conv2.bias, co, co)
        _wgrad(pl, xin, dz, ci, co, pl.s1, blk.conv1.weight, blk.conv1.bias, co, ci_real)
        if blk.downsample:
            _wgrad(pl, xin, dcur, ci, co, pl.sd, blk.conv_downsample.weight, blk.conv_downsample.bias, co, ci_real)
        if i == 0 and not need_dx0:
            break
        if blk.downsample:
            dtmp = pl.p(pl.dtmp, ci)
            _conv(pl, dcur, wv[(i, "dsT")], dtmp, co, ci, neg(pl.sd), pl.gx)
            addend = dtmp
        else:
            addend = dcur
        _conv(pl, dz, wv[(i, "c1T")], dprev, co, ci, neg(pl.s1), pl.gx, addend=addend)
    if need_dx0:
        pl.dout_entry = ops._call("scl_rs_copy", pl.p(pl.dx[0], pl.cps[0]), dx0.data_ptr(), pl.cps[0], 1, 1, ctypes.byref(pl.gx), S())
    pl.live_b, pl.bwd_need = _storages(d_out, dx0), need_dx0
    if record:
        pl.bwd_calls = ops.stop_recording()
    return dx0


class _ResStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, blocks, training, hold, *params):
        B, H, W, _ = x0.shape
        chans = [blocks[0].conv1.weight.shape[1]] + [b.conv1.weight.shape[0] for b in blocks]
        pl = _acquire(B, H, W, chans, x0.device, hold)
        out = _forward(pl, x0.contiguous().float(), blocks, training)
        _hold(ctx, pl)
        ctx.blocks, ctx.training = blocks, training
        return out

    @staticmethod
    def backward(ctx, d_out):
        pl = _owned(ctx)
        try:
            dx0 = _backward(pl, d_out, ctx.blocks, ctx.training, ctx.needs_input_grad[0])
        finally:
            pl.busy = False
        return (dx0, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 4)      # parameter gradients were accumulated into p.grad by the kernels


class _StackPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, blocks, attn, training, hold, *params):
        B, H, W, _ = x0.shape
        chans = [blocks[0].conv1.weight.shape[1]] + [b.conv1.weight.shape[0] for b in blocks]
        pl = _acquire(B, H, W, chans, x0.device, hold)
        e_S, e_T = _forward(pl, x0.contiguous().float(), blocks, training, attn)
        _hold(ctx, pl)
        ctx.blocks, ctx.training, ctx.attn = blocks, training, attn
        return e_S, e_T

    @staticmethod
    def backward(ctx, deS, deT):
        pl = _owned(ctx)
        try:
            dx0 = _backward(pl, (deS, deT), ctx.blocks, ctx.training, ctx.needs_input_grad[0], ctx.attn)
        finally:
            pl.busy = False
        return (dx0, None, None, None, None) + (None,) * (len(ctx.needs_input_grad) - 5)


def attn_supported(first_bn1, attention, pos_S, c_last):
    try:
        return (c_last == 64 and tuple(attention[0].weight.shape) == (128, 64, 1, 1) and tuple(attention[3].weight.shape) == (64, 128, 1, 1)
                and attention[2].weight.shape[0] == 128 and first_bn1.weight.shape[0] == 64 and pos_S.shape[-1] == 64)
    except (AttributeError, IndexError):
        return False


def res_stack_pool(x0, blocks, first_bn1, attention, pos_S):
    """The stack, first_bn1 + SELU, the attention block and the two attention poolings as one node:
    x0 [B, H, W, 1] -> (e_S [B, H, 64] (pos_S added), e_T [B, W, 64])."""
    blocks = list(blocks)
    training = bool(blocks[0].training)
    attn = (first_bn1, attention, pos_S)
    params = [p for b in blocks for p in b.parameters()] + list(first_bn1.parameters()) + list(attention.parameters()) + [pos_S]
    hold = torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in params))
    return _StackPoolFn.apply(x0, blocks, attn, training, hold, *params)


def res_stack(x0, blocks):
    """x0 [B, H, W, 1] -> [B, H, W, C]; `blocks`: the Residual_block modules (parameters, BatchNorm buffers and the training flag)."""
    blocks = list(blocks)
    training = bool(blocks[0].training)
    params = [p for b in blocks for p in b.parameters()]
    hold = torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in params))      # busy until the backward has run
    return _ResStackFn.apply(x0, blocks, training, hold, *params)
